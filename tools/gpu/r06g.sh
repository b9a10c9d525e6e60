O=gpurun_out/r06g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --no-launch-table --no-parity --no-selfcheck --steps 200 --warmup 30 --full-out $O/b.json > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/gap_analysis.py $f --launches 80 --steps 150 | tee $O/gap_analysis_c5_b4.txt
tail -1 $O/trace.log | cut -c1-200
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +6M -delete
