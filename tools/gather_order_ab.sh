#!/bin/bash
# Round 6: the deformable gather with the frame's queries walked in top-k order (query_order=0) against Morton order (1), same device, same
# call: kernel times by rocprofv3 --kernel-trace --stats (interleaved pairs), then the L1 / L2 counters of the gather in separate --pmc passes.
# usage: tools/gather_order_ab.sh <outdir under gpurun_out/> [times|pmc|all]
out=$1; what=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
B="python3 bench.py --batch 288 --streams 1 --no-graph --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 4 --warmup 2"
if [ $what != pmc ]; then
for rep in 1 2; do for q in 0 1; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t${q}_$rep -- $B --plan query_order=$q > $out/t${q}_$rep.log 2>&1 || exit 1
  f=$(find $out/t${q}_$rep -name "*kernel_stats.csv" | head -1)
  echo "order=$q rep=$rep: $(grep -m1 msda_raw $f | cut -d, -f1-6)"
done; done
fi
if [ $what != times ]; then
for q in 0 1; do
  mkdir -p $out/sum$q; : > $out/sum$q/gather.txt
  i=0
  for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
             "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TCP_GATE_EN1_sum"; do
    i=$((i+1))
    echo "order=$q pmc set $i: $set"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p${q}_$i -- $B --plan query_order=$q > $out/p${q}_$i.log 2>&1 || { echo "  (set $i failed: $(grep -m1 -i 'error\|exceeds' $out/p${q}_$i.log | cut -c1-160))"; continue; }
    python3 tools/pmc_summary.py $out/p${q}_$i | grep -A8 "== void moy::msda_raw" | grep -v "^== void moy::[a-ln-z]" >> $out/sum$q/gather.txt
  done
  echo "---- order=$q"; cat $out/sum$q/gather.txt
done
fi
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +4M -delete; find $out -name "*counter_collection.csv" -size +4M -delete; du -sh $out
