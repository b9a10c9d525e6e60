O=gpurun_out/r06x; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "topk" 2>&1 | tail -2 &&
python bench.py --config c5 --batch 4 --streams 1 --latency --steps 300 --warmup 30 --no-cpu-baseline --dump-launches $O/l_b4.json > $O/t.log 2>&1; echo "c5 b4 lat rc=$? $(tail -1 $O/t.log | cut -c60-150)"
python - <<'PY'
import json
L=json.load(open('gpurun_out/r06x/l_b4.json'))
print([ (x['name'], round(x['ms']*1e3,1)) for x in L if 'topk' in x['name']])
PY
python bench.py --batch 288 --streams 1 --no-cpu-baseline --no-live-traffic --steps 10 --warmup 3 --dump-launches $O/l_288.json > $O/t.log 2>&1; echo "b288 rc=$? $(tail -1 $O/t.log | cut -c60-150)"
python - <<'PY'
import json
L=json.load(open('gpurun_out/r06x/l_288.json'))
print([ (x['name'], round(x['ms']*1e3,1)) for x in L if 'topk' in x['name']])
PY
python -m pytest tests -x -q -m gpu -k "engine or temporal" 2>&1 | tail -2
