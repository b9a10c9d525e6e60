out=gpurun_out/r06q
mkdir -p $out
R=$(pwd)
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or split_f16" 2>&1 | tail -2 &&
for rep in 1 2; do for v in _prev ""; do
  MOYOLO_LIB=$R/mo_yolo_amd/libmoyolo$v.so python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 3 > $out/t.log 2>&1; echo "x3 lib$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
done; done
for v in _prev ""; do
  MOYOLO_LIB=$R/mo_yolo_amd/libmoyolo$v.so python bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 3 > $out/t.log 2>&1; echo "f32 lib$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
  MOYOLO_LIB=$R/mo_yolo_amd/libmoyolo$v.so python bench.py --config c5 --batch 4 --streams 1 --latency --steps 300 --warmup 30 --no-cpu-baseline > $out/t.log 2>&1; echo "c5 b4 lat lib$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
done
