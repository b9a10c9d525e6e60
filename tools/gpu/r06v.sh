O=gpurun_out/r06v; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python3 bench.py --dtype f32x3 --no-cpu-baseline --dump-launches $O/launches_c2_f32x3.json 2> $O/bench_c2_f32x3.log | tail -1 > $O/bench_c2_f32x3.json; cut -c1-200 $O/bench_c2_f32x3.json
python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$? $(tail -1 $O/bench_default.log | cut -c1-300)"
