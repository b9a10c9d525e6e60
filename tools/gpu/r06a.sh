set -x
mkdir -p gpurun_out/r06a
O=gpurun_out/r06a
python bench.py --config c5 --batch 4 --streams 1 --no-parity --no-selfcheck --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_c5_b4.json --full-out $O/bench_c5_b4.json > $O/c5_b4.log 2>&1
tail -1 $O/c5_b4.log | cut -c1-600
python bench.py --config c5 --batch 4 --streams 1 --no-graph --no-parity --no-selfcheck --no-cpu-baseline --no-launch-table --steps 300 --warmup 30 --full-out $O/bench_c5_b4_eager.json > $O/c5_b4_eager.log 2>&1
tail -1 $O/c5_b4_eager.log | cut -c1-300
python bench.py --config c2 --batch 1 --streams 1 --no-parity --no-selfcheck --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_c2_b1.json --full-out $O/bench_c2_b1.json > $O/c2_b1.log 2>&1
tail -1 $O/c2_b1.log | cut -c1-300
python bench.py --temporal 100 --batch 4 --no-parity --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_temporal_b4.json --full-out $O/bench_temporal_b4.json > $O/temporal_b4.log 2>&1
tail -1 $O/temporal_b4.log | cut -c1-300
python bench.py --dtype f32x3 --no-cpu-baseline --steps 10 --warmup 2 --dump-launches $O/launches_f32x3.json --full-out $O/bench_f32x3.json > $O/f32x3.log 2>&1
tail -1 $O/f32x3.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-launches $O/launches_default.json --full-out $O/bench_default.json > $O/default.log 2>&1
tail -1 $O/default.log | cut -c1-400
