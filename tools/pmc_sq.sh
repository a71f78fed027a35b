#!/usr/bin/env bash
# SQ-level counters of k_tick (instruction mix, LDS conflicts, wait cycles): separate --pmc passes, kernel-trace only.
# Usage on the GPU box:  bash tools/pmc_sq.sh <tag>     -> gpurun_out/pmc_sq_<tag>.txt
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS_ATOMIC"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$REPO/bench.py" --steps 100 --warmup 300 --no-cpu-baseline --no-copy-peak --no-verify --no-companion ${PMC_BENCH_ARGS:-} > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/pmc_sq_$TAG.txt"
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
    for row in rows:
        k = row.get("Kernel_Name", "")
        if "k_tick" not in k and "k_actor" not in k and "k_rollout" not in k:
            continue
        acc[k.split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
# mean/launch: launches of equal length (k_tick, chunked k_rollout); last: the run's LAST launch of the kernel = the timed
# --steps ticks of a persistent launch (its earlier launches are the prefill / warm-up calls, of other lengths)
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print("   %-28s mean/launch %.4g  last %.4g  (n=%d)" % (c, sum(v) / len(v), v[-1], len(v)))
PY
cat "$REPO/gpurun_out/pmc_sq_$TAG.txt"
