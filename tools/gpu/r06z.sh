O=gpurun_out/r06zz; mkdir -p $O
bash tools/profile_round.sh $O 2>&1 | tail -40
