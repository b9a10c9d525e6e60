out=gpurun_out/r06s
mkdir -p $out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "stem" 2>&1 | tail -3 &&
python3 tools/probes/stem_x3_time.py 2>&1 | grep stem
for rep in 1 2; do for v in 0 1; do
  python bench.py --plan stem_x3=$v --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 3 > $out/t.log 2>&1; echo "x3 stem_x3=$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
done; done
