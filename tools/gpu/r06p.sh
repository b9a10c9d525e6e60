O=gpurun_out/r06p; mkdir -p $O
R=$(pwd)
for rep in 1 2; do for v in "" _pin1 _pin2; do
  L=$R/mo_yolo_amd/libmoyolo$v.so
  MOYOLO_LIB=$L python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 3 --full-out $O/x3$v.json > $O/t.log 2>&1; echo "x3 lib$v rc=$? $(tail -1 $O/t.log | cut -c60-130)"
done; done
for v in "" _pin1 _pin2; do
  L=$R/mo_yolo_amd/libmoyolo$v.so
  MOYOLO_LIB=$L python bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 3 > $O/t.log 2>&1; echo "f32 lib$v rc=$? $(tail -1 $O/t.log | cut -c60-130)"
  MOYOLO_LIB=$L python bench.py --config c5 --batch 4 --streams 1 --latency --steps 300 --warmup 30 --no-cpu-baseline > $O/t.log 2>&1; echo "c5 b4 lat lib$v rc=$? $(tail -1 $O/t.log | cut -c60-130)"
done
