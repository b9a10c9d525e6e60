O=gpurun_out/r06x; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --full-out $O/bench_default.json > $O/default.log 2> $O/default.err; tail -1 $O/default.log | cut -c1-400; grep "^\[bench" $O/default.err | head -20
for kn in "MOY_NOP=1" "MOY_CONV_WS=2" "MOY_CONV_WS=2 MOY_GEMM_DMA=2"; do
  env $kn python bench.py --lab --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --no-parity --no-selfcheck --steps 300 --warmup 30 --dump-launches "$O/launches_c5_b4_$(echo $kn | tr ' =' '__').json" --full-out $O/b.json > $O/lat.log 2>&1
  echo "$kn: $(tail -1 $O/lat.log | cut -c60-150)"
done
