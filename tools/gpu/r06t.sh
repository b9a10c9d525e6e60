O=gpurun_out/r06t; mkdir -p $O
# the folded plan at the batch sizes between 13 and 104 frames (round 6: fold_min_rows 65536 -> 8192): parity gates of the bench at each size, then the whole GPU suite
for b in 13 16 24 48 96; do for pl in "fold_min_rows=65536" "fold_min_rows=8192"; do
  python bench.py --batch $b --streams 1 --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 --plan $pl --full-out $O/e_${b}_$pl.json > $O/t.log 2>&1; echo "B=$b $pl rc=$? $(tail -1 $O/t.log | cut -c60-130)"
done; done
python bench.py --config c5 --dtype f16 --batch 32 --streams 1 --no-cpu-baseline --no-live-traffic --steps 20 --warmup 3 > $O/t.log 2>&1; echo "c5 B=32 rc=$? $(tail -1 $O/t.log | cut -c60-130)"
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
