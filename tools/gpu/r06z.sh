O=gpurun_out/r06z; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
bash tools/profile_round.sh $O 2>&1 | tail -40
