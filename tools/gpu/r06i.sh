O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "decoder_tail or gemm_wreg or wreg or score" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
python -m pytest tests/test_gpu_engine.py tests/test_gpu_temporal.py tests/test_gpu_modules.py -x -q > $O/t_eng.log 2>&1; tail -3 $O/t_eng.log
python bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_c5_b4.json --full-out $O/bench_c5_b4_lat.json > $O/c5_b4_lat.log 2>&1
tail -1 $O/c5_b4_lat.log | cut -c1-300
python bench.py --temporal 100 --batch 4 --latency --no-cpu-baseline --steps 300 --warmup 30 --full-out $O/bench_temporal_b4_lat.json > $O/temporal_b4_lat.log 2>&1
tail -1 $O/temporal_b4_lat.log | cut -c1-300
python bench.py --config c2 --batch 1 --streams 1 --latency --no-cpu-baseline --no-parity --steps 300 --warmup 30 --full-out $O/bench_c2_b1_lat.json > $O/c2_b1_lat.log 2>&1
tail -1 $O/c2_b1_lat.log | cut -c1-300
