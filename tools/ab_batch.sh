# usage: bash tools/ab_batch.sh "<batch>x<streams> ..." [bench args]   -- bench throughput by frames per step and number of streams (same device)
mkdir -p gpurun_out/ab
cfgs=$1; shift
for cfg in $cfgs; do
  b=${cfg%x*}; s=${cfg#*x}
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --batch $b --streams $s --no-cpu-baseline --no-selfcheck --no-parity --no-launch-table --full-out gpurun_out/ab/full_$cfg.json "$@" > gpurun_out/ab/b_$cfg.json 2> gpurun_out/ab/b_$cfg.err || { echo "$cfg failed"; tail -2 gpurun_out/ab/b_$cfg.err; continue; }
  python - <<P
import json
d=json.loads(open("gpurun_out/ab/b_$cfg.json").read().strip().splitlines()[-1])
print("$cfg $*:", d["value"], "frames/s", d["ms_per_step"], "ms/step", d["config"].get("hbm_allocated_gb"), "GB", flush=True)
P
done
