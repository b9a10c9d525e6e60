# usage: bash tools/ab_env.sh NAME=VALUE [reps]   -- interleaved bench runs with and without one environment switch (same device)
set -e
kv="$1"; reps="${2:-2}"
mkdir -p gpurun_out/ab
for i in $(seq 1 $reps); do
for v in off on; do
  if [ $v = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
  timeout -k 10 200 python bench.py --lab --steps 20 --warmup 3 --no-cpu-baseline --no-selfcheck --full-out gpurun_out/ab/full_$v.json > gpurun_out/ab/b_$v.json 2> gpurun_out/ab/b_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/ab/b_$v.json").read().strip().splitlines()[-1])
print("$kv $v run $i:", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:34], d["roofline"]["avg_ms"], d["parity"]["ok"], flush=True)
P
done
done
