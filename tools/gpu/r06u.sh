O=gpurun_out/r06u; mkdir -p $O
python bench.py --gpus 6 --rehearse-one-gpu --batch 32 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline --full-out $O/bench_6ranks_one_gpu.json > $O/r6.log 2>&1; echo "6 ranks rc=$? $(tail -1 $O/r6.log | cut -c1-200)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --batch 64 --streams 1 --steps 5 --warmup 2 --full-out $O/bench_torchrun_2ranks_one_gpu.json > $O/t2.log 2>&1; echo "torchrun 2 ranks rc=$? $(tail -1 $O/t2.log | cut -c1-200)"
