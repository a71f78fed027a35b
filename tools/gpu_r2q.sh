#!/bin/bash
set -u
export TMPDIR=/tmp
REPO=$(pwd)
O=$REPO/gpurun_out/r2q
mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $O/p1 -o p1 --output-format csv -- python3 $REPO/bench.py --actor --steps 50 --warmup 0 --no-cpu-baseline --no-copy-peak --pipeline 1 --mode step > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/p2 -o p2 --output-format csv -- python3 $REPO/bench.py --actor --steps 50 --warmup 0 --no-cpu-baseline --no-copy-peak --pipeline 1 --mode step > $O/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/st -o r -- python3 $REPO/bench.py --actor --steps 100 --warmup 0 --no-cpu-baseline --no-copy-peak --pipeline 1 --mode step > $O/st.log 2>&1
cd $REPO
python3 - $O <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_actor" not in k: continue
        acc[k.split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c][-30:]
        print("   %-32s mean/launch %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
python tools/rocprof_summary.py $O/st/r_results.db --tail 50 | head -8
