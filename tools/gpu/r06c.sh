O=gpurun_out/r06c; mkdir -p $O
python tools/probes/raw_f32_debug.py > $O/dbg.log 2>&1; cat $O/dbg.log
