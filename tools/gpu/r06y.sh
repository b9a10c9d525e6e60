O=gpurun_out/r06y; mkdir -p $O
python bench.py --extra-legs --full-out $O/bench_full.json > $O/bench_extra.log 2> $O/bench_extra.err; echo "rc=$?"; grep -c '"leg"' $O/bench_extra.log; tail -1 $O/bench_extra.log | cut -c1-300
