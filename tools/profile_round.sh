#!/bin/bash
# Everything the round's profiles/ entries come from, in one GPU call.  usage: tools/profile_round.sh <outdir under gpurun_out/>
#   1. bench JSON lines: default (C2 bf16, with the CPU baseline), --dtype f16, --config c4, --config c5, --temporal 100
#   2. rocprofv3 --kernel-trace --stats of the default run (two streams) and of one sub-batch engine on one stream
#   3. HBM traffic of the whole plan: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md §HBM), one engine,
#      one stream, eager launches (counter collection serialises the dispatches); 5 passes of the plan = build pass + first forward + 1 warm-up + 2 steps
# Steps are joined with && : a step that fails or times out ends the call.   Second argument: "lines" (1 only), "prof" (2 + 3), default both.
set -o pipefail
out=$1; stage=${2:-all}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
B="python3 bench.py"
run_line() {   # name, args...
  local name=$1; shift
  echo "[profile_round] bench $name: $*"
  timeout -k 10 600 $B "$@" 2> $out/bench_$name.log | tail -1 > $out/bench_$name.json && cut -c1-300 $out/bench_$name.json
}
if [ "$stage" != "prof" ]; then
run_line c2_bf16 --dump-launches $out/launches_c2_bf16.json &&
run_line c2_f16 --dtype f16 --no-cpu-baseline --no-live-traffic --dump-launches $out/launches_c2_f16.json &&
run_line c4_bf16 --config c4 --no-cpu-baseline --dump-launches $out/launches_c4_bf16.json &&
run_line c5_f16 --config c5 --no-cpu-baseline --no-live-traffic &&
run_line c2_temporal100 --temporal 100 --batch 32 --no-cpu-baseline &&
run_line c2_bf16_1stream --batch 288 --streams 1 --no-cpu-baseline --no-live-traffic &&
run_line c5_f16_b4_latency --config c5 --batch 4 --streams 1 --latency --steps 300 --warmup 30 --no-cpu-baseline --dump-launches $out/launches_c5_f16_b4_latency.json &&
run_line c2_bf16_temporal100_b4_latency --temporal 100 --batch 4 --latency --steps 300 --warmup 30 --no-cpu-baseline &&
run_line c2_bf16_b1_latency --config c2 --batch 1 --streams 1 --latency --steps 300 --warmup 30 --no-cpu-baseline --no-parity &&
run_line c2_f32 --dtype f32 --no-cpu-baseline --dump-launches $out/launches_c2_f32.json &&
run_line c2_f32x3 --dtype f32x3 --no-cpu-baseline --dump-launches $out/launches_c2_f32x3.json &&
run_line full_bf16 --config full --no-cpu-baseline --dump-launches $out/launches_full_bf16.json || exit 1
fi
[ "$stage" = "lines" ] && exit 0
echo "[profile_round] rocprofv3 kernel stats (default run, 2 streams)" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_default -- python3 bench.py --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 10 --warmup 2 > $out/trace_default.log 2>&1 &&
echo "[profile_round] rocprofv3 kernel stats (288 frames, 1 stream)" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_1stream -- python3 bench.py --batch 288 --streams 1 --no-cpu-baseline --no-parity --no-selfcheck --no-live-traffic --steps 10 --warmup 2 > $out/trace_1stream.log 2>&1 &&
echo "[profile_round] PMC FETCH_SIZE" &&
timeout -k 10 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --batch 288 --streams 1 --no-graph --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 2 --warmup 1 > $out/fetch.log 2>&1 &&
echo "[profile_round] PMC WRITE_SIZE" &&
timeout -k 10 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --batch 288 --streams 1 --no-graph --no-cpu-baseline --no-parity --no-launch-table --no-selfcheck --steps 2 --warmup 1 > $out/write.log 2>&1 &&
python3 tools/traffic_summary.py $out --steps 5 --frames 288 | tail -20 &&
python3 tools/trace_by_grid.py $out/trace_1stream > $out/trace_1stream_by_launch_shape.csv &&
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +8M -delete; find $out -name "*counter_collection.csv" -size +8M -delete; du -sh $out
