#!/usr/bin/env bash
# Texture-addresser / vector-L1 counters of the headline command (is the address unit busy with the 16-byte row stores?).
# Usage on the GPU box:  bash tools/pmc_ta.sh <tag> [bench args]   -> gpurun_out/pmc_ta_<tag>.txt
set -u
TAG=${1:-ta}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_ta_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/p$i" -o p$i --output-format csv -- python3 "$REPO/bench.py" --steps 100 --warmup 300 --no-cpu-baseline --no-copy-peak --no-verify --no-companion "$@" > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/pmc_ta_$TAG.txt"
import sys, glob, csv, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0)))
    for row in rows:
        k = row.get("Kernel_Name", "")
        if "k_tick" not in k and "k_rollout" not in k:
            continue
        acc[k.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print("   %-36s mean/launch %.4g  last %.4g  (n=%d)" % (c, sum(v) / len(v), v[-1], len(v)))
PY
rm -rf "$OUT"/p*/
cat "$REPO/gpurun_out/pmc_ta_$TAG.txt"
