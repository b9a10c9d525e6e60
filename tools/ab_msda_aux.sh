set -e
for i in 1 2; do
for x in d1 a1 a2 a3 a17; do
  MOYOLO_LIB=$PWD/ab_libs/libmoyolo_$x.so timeout -k 10 200 python tools/probes/msda_raw_time.py | sed "s/^/$x: /"
done
done
