O=gpurun_out/r06w; mkdir -p $O
python tools/parity_stream.py --config c2 --frames 600 --seqs 0 1 --batch 120 --x3 --out $O/parity_r06_c2_2x600_frames_b120_with_f32x3.json > $O/parity.log 2>&1; echo "rc=$? $(tail -1 $O/parity.log | cut -c1-600)"
