set -e
mkdir -p gpurun_out/r05m
for i in 1 2; do
for v in 0 1; do
  MOY_MR_MFMA=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-selfcheck --dump-launches gpurun_out/r05m/l_$v.json --full-out gpurun_out/r05m/full_$v.json > gpurun_out/r05m/b_${v}_$i.json 2> gpurun_out/r05m/b_${v}_$i.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r05m/b_${v}_$i.json").read().strip().splitlines()[-1])
print("MOY_MR_MFMA=$v run $i:", d["value"], d["ms_per_step"], d["roofline"]["kernel"][:30], d["roofline"]["avg_ms"], d["parity"]["ok"], flush=True)
P
done
done
