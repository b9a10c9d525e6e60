O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_gpu_engine.py -x -q > $O/t_engine.log 2>&1; tail -5 $O/t_engine.log
python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 2 --dump-launches $O/launches_f32x3.json --full-out $O/bench_f32x3.json > $O/f32x3.log 2>&1
tail -1 $O/f32x3.log | cut -c1-300
python bench.py --dtype f32x3 --no-cpu-baseline --no-launch-table --no-parity --no-selfcheck --steps 10 --warmup 2 --batch 192 --streams 4 --full-out $O/bench_f32x3_4x48.json > $O/f32x3_4x48.log 2>&1
tail -1 $O/f32x3_4x48.log | cut -c1-200
python bench.py --dtype f32x3 --no-cpu-baseline --no-launch-table --no-parity --no-selfcheck --steps 10 --warmup 2 --batch 384 --streams 4 --full-out $O/bench_f32x3_4x96.json > $O/f32x3_4x96.log 2>&1
tail -1 $O/f32x3_4x96.log | cut -c1-200
python bench.py --dtype f32x3 --no-cpu-baseline --no-launch-table --no-parity --no-selfcheck --steps 10 --warmup 2 --batch 288 --streams 2 --full-out $O/bench_f32x3_2x144.json > $O/f32x3_2x144.log 2>&1
tail -1 $O/f32x3_2x144.log | cut -c1-200
python bench.py --dtype f32 --no-cpu-baseline --steps 10 --warmup 2 --dump-launches $O/launches_f32.json --full-out $O/bench_f32.json > $O/f32.log 2>&1
tail -1 $O/f32.log | cut -c1-300
python bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --steps 300 --warmup 30 --full-out $O/bench_c5_b4_lat.json > $O/c5_b4_lat.log 2>&1
tail -1 $O/c5_b4_lat.log | cut -c1-1800
python bench.py --steps 20 --warmup 5 --dump-launches $O/launches_default.json --full-out $O/bench_default.json > $O/default.log 2>&1
tail -1 $O/default.log | cut -c1-3000
