#!/bin/bash
# Same-device A/B of the GEMM micro-benchmark: mo_yolo_amd/libmoyolo_base.so (A) vs mo_yolo_amd/libmoyolo.so (B)
export BG_B=${BG_B:-64} BG_ROUNDS=${BG_ROUNDS:-3}
for i in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export MOYOLO_LIB=$PWD/mo_yolo_amd/libmoyolo_base.so; else unset MOYOLO_LIB; fi
    echo "== $v"; timeout 300 python tools/bench_gemm.py 2>&1 | grep -v "amdgpu.ids" | awk '{printf "%s %s %s | ", $1, $2, $(NF-5)} END {print ""}'
  done
done
