O=gpurun_out/r06u; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --rehearse-one-gpu --batch 64 --streams 1 --steps 5 --warmup 2 --full-out $O/bench_torchrun_2ranks_one_gpu.json > $O/torchrun.log 2> $O/torchrun.err
echo "rc=$?"; tail -1 $O/torchrun.log | cut -c1-1500; grep -c "Traceback" $O/torchrun.err
