set -e
mkdir -p gpurun_out/r05m
for i in 1 2; do
for d in 1 2 3; do
  MOYOLO_LIB=$PWD/ab_libs/libmoyolo_d$d.so timeout -k 10 200 python tools/probes/msda_raw_time.py | sed "s/^/depth $d: /"
done
done
for d in 1 2 3; do
  MOYOLO_LIB=$PWD/ab_libs/libmoyolo_d$d.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-selfcheck --full-out gpurun_out/r05m/full_d$d.json > gpurun_out/r05m/b_d$d.json 2> gpurun_out/r05m/b_d$d.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r05m/b_d$d.json").read().strip().splitlines()[-1])
print("depth $d:", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:30], d["roofline"]["avg_ms"], d["parity"]["ok"], flush=True)
P
done
