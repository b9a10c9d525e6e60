O=gpurun_out/r06e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
B="python3 bench.py --lab --batch 288 --streams 1 --no-graph --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 4 --warmup 2"
for rep in 1 2; do for x4 in 0 1; do for q in 0 1; do
  export MOY_MRM_X4=$x4 MOY_Q_ORDER=$q
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_${x4}_${q}_$rep -- $B > $O/t_${x4}_${q}_$rep.log 2>&1 || exit 1
  f=$(find $O/t_${x4}_${q}_$rep -name "*kernel_stats.csv" | head -1)
  echo "x4=$x4 order=$q rep=$rep: $(grep -m1 msda_raw $f | cut -d, -f2-6)"
done; done; done
unset MOY_MRM_X4 MOY_Q_ORDER
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
python bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --steps 300 --warmup 30 --dump-launches $O/launches_c5_b4.json --full-out $O/bench_c5_b4_lat.json > $O/c5_b4_lat.log 2>&1
tail -1 $O/c5_b4_lat.log | cut -c1-900
python bench.py --temporal 100 --batch 4 --latency --no-cpu-baseline --steps 300 --warmup 30 --full-out $O/bench_temporal_b4_lat.json > $O/temporal_b4_lat.log 2>&1
tail -1 $O/temporal_b4_lat.log | cut -c1-900
