#!/usr/bin/env bash
# Memory-path stall counters of k_tick (TA / TCP / TCC write side, UTCL1): separate --pmc passes, kernel-trace only.
# (a TA_* group aborted rocprofv3 on this pool and hung in finalisation: not collected; every pass is under `timeout`)
# Usage on the GPU box:  bash tools/pmc_mem.sh <tag>     -> gpurun_out/pmc_mem_<tag>.txt
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_mem_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_WRITE_TAGCONFLICT_STALL_CYCLES" \
           "TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS" \
           "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL" \
           "TCC_BUSY TCC_TAG_STALL TCC_WRITE TCC_WRITEBACK" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$REPO/bench.py" --steps 20 --warmup 300 --no-cpu-baseline ${PMC_BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/pmc_mem_$TAG.txt"
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_tick" not in k:
            continue
        acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c][-20:]
        print("   %-36s mean/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat "$REPO/gpurun_out/pmc_mem_$TAG.txt"
