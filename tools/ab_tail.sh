set -e
for i in 1 2; do
for x in base pd1 pd2 pd3; do
  MOYOLO_LIB=$PWD/ab_libs/libmoyolo_$x.so timeout -k 10 200 python tools/probes/tail_diag.py | sed "s/^/$x: /"
done
done
MOYOLO_LIB=$PWD/ab_libs/libmoyolo_pd3.so MOY_TAIL_ABL=2 timeout -k 10 200 python tools/probes/tail_diag.py 2>/dev/null | tail -17
