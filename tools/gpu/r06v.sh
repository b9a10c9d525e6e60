O=gpurun_out/r06v; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_stress.py -x -q -k "conv or stress or ring" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-launches $O/launches_default.json --full-out $O/bench_default.json > $O/default.log 2> $O/default.err; tail -1 $O/default.log | cut -c1-200
python - <<'P'
import json
d=json.load(open('gpurun_out/r06v/launches_default.json'))
print(round(sum(x['ms'] for x in d),3))
for x in d:
    if 'M186048 N128 K1152' in x['name']: print(x['ms'], x['name'])
P
