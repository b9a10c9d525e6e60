O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_c5_b4.json --full-out $O/bench_c5_b4_lat.json > $O/c5_b4_lat.log 2>&1
tail -1 $O/c5_b4_lat.log | cut -c1-700
python bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --no-launch-table --no-parity --no-selfcheck --steps 300 --warmup 30 --plan fork_small_value=0 --full-out $O/bench_c5_b4_lat_nofork.json > $O/c5_b4_lat_nofork.log 2>&1
tail -1 $O/c5_b4_lat_nofork.log | cut -c1-300
python bench.py --temporal 100 --batch 4 --latency --no-cpu-baseline --steps 300 --warmup 30 --full-out $O/bench_temporal_b4_lat.json > $O/temporal_b4_lat.log 2>&1
tail -1 $O/temporal_b4_lat.log | cut -c1-400
python bench.py --steps 20 --warmup 5 --dump-launches $O/launches_default.json --full-out $O/bench_default.json > $O/default.log 2>&1
tail -1 $O/default.log | cut -c1-1200
python bench.py --gpus 6 --rehearse-one-gpu --batch 32 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --no-launch-table --full-out $O/bench_rehearsal_6ranks_one_gpu.json > $O/rehearsal.log 2>&1
tail -1 $O/rehearsal.log | cut -c1-2500
