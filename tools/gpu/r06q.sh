out=gpurun_out/r06q
mkdir -p $out
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split_f16" 2>&1 | tail -3 &&
for rep in 1 2; do for v in 0 1; do
  MOY_X3_LN128=$v python bench.py --lab --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 3 > $out/t.log 2>&1; echo "x3 ln128=$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
done; done
python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 3 --dump-launches $out/launches_x3.json > $out/t.log 2>&1; echo "x3 rc=$? $(tail -1 $out/t.log | cut -c60-130)"
python -m pytest tests -x -q -m gpu -k "x3 or f32x3 or split" 2>&1 | tail -3
