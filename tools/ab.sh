#!/bin/bash
# Same-device A/B of two builds of libmoyolo.so (devices of the pool differ by up to ~10 %: never compare across gpurun calls).
# usage: tools/ab.sh [bench.py args...]   with mo_yolo_amd/libmoyolo_base.so = build A, mo_yolo_amd/libmoyolo.so = build B
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export MOYOLO_LIB=$PWD/mo_yolo_amd/libmoyolo_base.so; else unset MOYOLO_LIB; fi
    echo -n "$v: "; timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>&1 | grep metric | sed 's/.*"value": \([0-9.]*\).*/\1/'
  done
done
