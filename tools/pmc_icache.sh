#!/usr/bin/env bash
# instruction-cache / issue-stall counters of the tick kernels (separate --pmc passes, kernel-trace only)
# Usage on the GPU box:  bash tools/pmc_icache.sh <tag> [bench args]   -> gpurun_out/pmc_icache_<tag>.txt
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_icache_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$REPO/bench.py" --steps 100 --warmup 300 --no-cpu-baseline --no-copy-peak "$@" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/pmc_icache_$TAG.txt"
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_tick" not in k and "k_actor" not in k and "k_rollout" not in k:
            continue
        acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print("   %-28s mean/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
cat "$REPO/gpurun_out/pmc_icache_$TAG.txt"
