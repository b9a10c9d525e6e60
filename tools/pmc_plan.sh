#!/bin/bash
# SQ counters of EVERY kernel of the plan (one engine, 288 frames, one stream, eager launches), three --pmc passes; summary per kernel.
# usage: tools/pmc_plan.sh <outdir under gpurun_out/> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
B="python3 bench.py --batch 288 --streams 1 --no-graph --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p1 -- $B "$@" > $out/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p2 -- $B "$@" > $out/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT --output-format csv -d $out/p3 -- $B "$@" > $out/p3.log 2>&1 &&
python3 tools/pmc_summary.py $out > $out/summary_all.txt && grep -A24 "== void moy::" $out/summary_all.txt > $out/summary.txt
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +8M -delete; find $out -name "*counter_collection.csv" -size +8M -delete; du -sh $out
