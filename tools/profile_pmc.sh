#!/bin/bash
# SQ / TCP counter passes for the bench workload (each --pmc set is its own run). Usage: tools/profile_pmc.sh [tag]
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $OUT/$name.log 2>&1; }
run sqA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sqB SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run tcpA TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run tcpB TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run tccA TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
run tccB TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum
python3 - <<PY
import csv, glob, collections, re
def short(name):
    m = re.search(r"(tracePacketKernel<\w+, \w+>|tracePacketKernel<\w+>|traceKernel<\w+>|shadeKernel<\w+, \w+>|generateKernel|renderReferenceOrder)", name)
    return m.group(1) if m else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k and not k.endswith("true>"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k in sorted(agg):
        out.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]; out.write(f"   {c:42s} n={len(v):3d} avg={sum(v)/len(v):16.1f}\n")
print(open("$OUT/summary.txt").read())
PY
