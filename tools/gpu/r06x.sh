python3 tools/probes/small_batch_streams.py 2>&1 | grep -v amdgpu.ids | tail -5
SB_B=8 python3 tools/probes/small_batch_streams.py 2>&1 | grep -v amdgpu.ids | tail -5
