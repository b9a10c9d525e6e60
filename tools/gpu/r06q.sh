out=gpurun_out/r06q
mkdir -p $out
R=$(pwd)
# timing-only: every product of the split form into ONE accumulator (wrong scale), two stages in flight for the 64 x 64 wave tiles
for rep in 1 2; do for v in "" _x3s; do
  MOYOLO_LIB=$R/mo_yolo_amd/libmoyolo$v.so python bench.py --dtype f32x3 --no-cpu-baseline --no-parity --no-selfcheck --steps 10 --warmup 3 > $out/t.log 2>&1; echo "x3 lib$v rc=$? $(tail -1 $out/t.log | cut -c60-130)"
done; done
export BG_DT=f32x3 BG_B=96 BG_ONLY="L1 ,L4m,L2cv2,L3 ,L4cv2,L6m 3x3,L5 ,value,L2m"
for v in "" _x3s; do echo "== lib$v"; MOYOLO_LIB=$R/mo_yolo_amd/libmoyolo$v.so BG_ROUNDS=3 BG_REPS=5 python3 tools/bench_gemm.py 2>&1 | tail -10; done
