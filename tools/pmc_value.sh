#!/bin/bash
# PMC counters of the value projection (tools/probes/wreg_ablate.py in its single-process child mode: the kernel, then torch's fill of
# the same 12 GB and a 2 GB copy for comparison) -- the write path: TCC -> EA write requests and their stalls, TA / TCP stalls, SQ waits.
# Separate --pmc passes (MI355X_MICROARCH.md); the program itself after "--".
out=${PMC_OUT:-gpurun_out/pmc_value}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $root
mkdir -p $out
export WREG_ABLATE_CHILD=1 MOY_WREG_ABL=0
run() {   # pass name, counters...
  local name=$1; shift
  timeout -k 10 90 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 tools/probes/wreg_ablate.py 288 > $out/$name.log 2>&1
}
# (at most four counters of one block per pass: a larger request "exceeds the capabilities of the hardware to collect")
# one failing pass must not cost the others their data: the passes are independent processes, each bounded by its own timeout
run p1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_CYCLE_sum &&
run p2 TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum &&
run p7 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU &&
run p6 TCC_IB_STALL_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum &&
run p3 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum &&
run p4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum &&
run p5 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
python3 - <<PY
import csv, glob, collections
# per kernel class: counters summed over its dispatches / number of calls (the 12 GB fill is 8 dispatches per call)
for pdir in ("p1", "p2", "p3", "p4", "p5", "p6", "p7"):
    d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$out/" + pdir + "/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            nm = r["Kernel_Name"]
            key = "value projection: 12 GB written + 2 GB read" if "gemm_wreg_kernel" in nm else ("torch fill, 12 GB written" if "FillFunctor<c10::BFloat16>" in nm else None)
            if not key: continue
            d[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if (key, r["Dispatch_Id"]) not in seen: seen.add((key, r["Dispatch_Id"])); n[key] += 1
    for k, c in d.items():
        calls = n[k] / (8 if "fill" in k else 1)
        print(f"{pdir}  {k}  ({n[k]} dispatches = {calls:.0f} calls; per call:)")
        for name, v in sorted(c.items()): print(f"      {name:44s} {v / calls:16.0f}")
PY
