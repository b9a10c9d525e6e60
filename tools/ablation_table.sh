#!/bin/bash
# Same-device ablation of the kernels added in round 2: the default plan, then each knob switched off alone (bench.py, C2, bf16,
# 576 frames/step; no CPU baseline).  usage: tools/ablation_table.sh <outfile>
out=${1:-gpurun_out/ablation.txt}
mkdir -p $(dirname $out)
: > $out
run() {   # label, env assignment (or "")
  local label=$1 kv=$2
  local line
  line=$(env $kv python3 bench.py --lab --no-cpu-baseline --no-launch-table 2>/dev/null | tail -1) || return 1
  python3 - "$label" "$kv" "$line" >> $out <<'PY'
import json, sys
label, kv, line = sys.argv[1:4]
d = json.loads(line)
print(f"{label:58s} {kv:22s} {d['value']:9.1f} FPS  {d['ms_per_step']:7.3f} ms/step  parity ok {d['parity']['ok'] if d.get('parity') else None}")
PY
  tail -1 $out
}
run "default plan" "MOY_NOP=1" &&
run "stem + layer 1 as two launches" "MOY_FUSE_STEM=0" &&
run "first C2f as four launches" "MOY_FUSE_C2F=0" &&
run "3x3 s1 convs on the tile-per-block direct kernel" "MOY_CONV_WS=0" &&
run "3x3 s2 convs on the implicit-GEMM kernel" "MOY_CONV_S2=0" &&
run "deformable gather: one corner per load" "MOY_MSDA_PLANES=0" &&
run "1x1 convs with 128 outputs on the tiled kernel" "MOY_WREG_N128=0" &&
run "all linears / 1x1 convs on the tiled kernel" "MOY_GEMM_WREG=0" &&
run "default plan (again)" "MOY_NOP=1"
