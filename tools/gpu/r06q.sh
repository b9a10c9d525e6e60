out=gpurun_out/r06q2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $out; mkdir -p $out
export BG_DT=f32x3 BG_B=96 BG_ONLY="L4m,L6m 3x3,L4cv2,value,L3 " BG_ROUNDS=1 BG_REPS=2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p1 -- python3 tools/bench_gemm.py > $out/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p2 -- python3 tools/bench_gemm.py > $out/p2.log 2>&1
python3 tools/pmc_summary.py $out > $out/summary.txt; find $out -name "*.db" -delete; tail -8 $out/p2.log | grep -v rocprof | head -8
