O=gpurun_out/r06x; mkdir -p $O
for i in 1 2 3; do python bench.py --no-cpu-baseline > $O/b$i.log 2>&1; echo "run $i rc=$? $(tail -1 $O/b$i.log | cut -c60-140)"; done
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -2
