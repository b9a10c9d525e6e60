#!/bin/bash
# HBM traffic of the bench's kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots),
# as MI355X_MICROARCH.md "HBM" prescribes.  usage: tools/pmc_traffic.sh <outdir> [bench args]
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --no-graph --no-cpu-baseline --steps 2 --warmup 1 "$@" > $out/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 "$@" > $out/trace.log 2>&1
