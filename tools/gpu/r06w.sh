O=gpurun_out/r06w; mkdir -p $O
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or conv" > $O/t_ops.log 2>&1; tail -2 $O/t_ops.log
for rep in 1 2; do for kn in "MOY_SMALL_NSET=2" "MOY_SMALL_NSET=4" "MOY_SMALL_NSET=4 MOY_SMALL_NSET_BLOCKS=1024"; do
  env $kn python bench.py --lab --config c5 --batch 4 --streams 1 --latency --no-cpu-baseline --no-parity --no-selfcheck --steps 300 --warmup 30 --dump-launches "$O/launches_c5_b4_$(echo $kn | tr ' =' '__').json" --full-out $O/b.json > $O/lat.log 2>&1
  echo "$kn rep $rep: $(tail -1 $O/lat.log | cut -c60-150)"
done; done
