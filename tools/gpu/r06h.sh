O=gpurun_out/r06h; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "split_f16 or gemm or conv" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
for rep in 1 2; do for ns in 2 4; do
  MOY_X3_NSET=$ns python bench.py --lab --dtype f32x3 --no-cpu-baseline --no-parity --no-selfcheck --steps 10 --warmup 2 --dump-launches $O/launches_f32x3_nset$ns.json --full-out $O/bench_f32x3_nset$ns.json > $O/f32x3_nset$ns.log 2>&1
  echo "nset=$ns rep=$rep: $(tail -1 $O/f32x3_nset$ns.log | cut -c60-140)"
done; done
python -m pytest tests/test_gpu_engine.py -x -q -k "split_f16 or fp32 or f32" > $O/t_eng.log 2>&1; tail -2 $O/t_eng.log
python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 2 --full-out $O/bench_f32x3.json > $O/f32x3.log 2>&1
tail -1 $O/f32x3.log | cut -c1-1800
