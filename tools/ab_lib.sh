# usage: bash tools/ab_lib.sh <other libmoyolo.so> [reps]   -- interleaved bench runs: the in-tree library and another build (same device)
set -e
other="$1"; reps="${2:-2}"
mkdir -p gpurun_out/ab
for i in $(seq 1 $reps); do
for v in other tree; do
  if [ $v = other ]; then export MOYOLO_LIB="$other" MOYOLO_LIB_ALLOW_MISSING=1; else unset MOYOLO_LIB MOYOLO_LIB_ALLOW_MISSING; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-selfcheck --dump-launches gpurun_out/ab/l_$v.json --full-out gpurun_out/ab/full_$v.json > gpurun_out/ab/b_$v.json 2> gpurun_out/ab/b_$v.err
  python - <<P
import json
d=json.loads(open("gpurun_out/ab/b_$v.json").read().strip().splitlines()[-1])
l=json.load(open("gpurun_out/ab/l_$v.json"))
mha=[r["ms"] for r in l if r["name"].startswith("mha_core")]
print("$v run $i:", d["value"], d["ms_per_step"], "mha avg ms", round(sum(mha)/max(len(mha),1),4), d["parity"]["ok"], flush=True)
P
done
done
