#!/bin/bash
out=${PMC_OUT:-gpurun_out/pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
export BG_B=${BG_B:-96} BG_ONLY="${BG_ONLY:-value,L1 ,L4m,L6m}" BG_ROUNDS=1 BG_REPS=3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p1 -- python3 tools/bench_gemm.py > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p2 -- python3 tools/bench_gemm.py > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS --output-format csv -d $out/p3 -- python3 tools/bench_gemm.py > $out/p3.log 2>&1
python3 tools/pmc_summary.py $out | head -400 > $out/summary.txt
tail -5 $out/p3.log
