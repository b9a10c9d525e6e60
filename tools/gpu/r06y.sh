O=gpurun_out/r06y; mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --extra-legs --full-out $O/bench_full.json > $O/default_with_extra_legs.jsonl 2> $O/default_with_extra_legs.err
grep -c "^{" $O/default_with_extra_legs.jsonl; cut -c1-260 $O/default_with_extra_legs.jsonl
tail -3 $O/default_with_extra_legs.err
