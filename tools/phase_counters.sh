#!/usr/bin/env bash
# Per-phase hardware counters of the 12-lane tick: k_tick<128> truncated behind phase n (pve_debug_stop_phase) on one frozen
# steady-state batch of 2048 x 128; counters(n) - counters(n - 1) = phase n.  Separate --pmc passes, kernel-trace only.
# Usage on the GPU box:  bash tools/phase_counters.sh     -> gpurun_out/phase_counters.txt
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/phase_counters
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for stop in 0 1 2 3 4 5 6 7 8 -1; do
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $grp --kernel-trace -d "$OUT/s${stop}_p$i" -o r --output-format csv -- python3 "$REPO/tools/phase_probe.py" $stop > "$OUT/s${stop}_p$i.log" 2>&1
  done
done
python3 - "$OUT" <<'PY' > "$REPO/gpurun_out/phase_counters.txt"
import sys, glob, csv, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/s*_p*/**/*counter_collection.csv", recursive=True):
    stop = int(re.search(r"/s(-?\d+)_p", f).group(1))
    for row in csv.DictReader(open(f)):
        if "k_tick" in row.get("Kernel_Name", ""):
            acc[stop][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = ["load", "step1", "step2+listsA", "step3+listsB", "build", "rank", "walk+reward", "effects", "lock+lock2", "final"]
order = [0, 1, 2, 3, 4, 5, 6, 7, 8, -1]
ctrs = sorted({c for s in acc for c in acc[s]})
waves = 2048 * 2.0
print("per wave and tick of k_tick<128> (2048 envs x 128 slots, steady state ~85 alive / 50 controlled), phase = difference of truncated launches")
print("%-14s" % "phase" + "".join("%22s" % c for c in ctrs))
prev = {c: 0.0 for c in ctrs}
for s, nm in zip(order, names):
    # the counted launches are every second one of the run's 40 k_tick launches (full tick, truncated tick, ...; the 400
    # prefill ticks run in k_rollout)
    cur = {c: (sum(acc[s][c][-40:][1::2]) / max(1, len(acc[s][c][-40:][1::2]))) for c in ctrs}
    print("%-14s" % nm + "".join("%22.1f" % ((cur[c] - prev[c]) / waves) for c in ctrs))
    prev = cur
print("%-14s" % "whole tick" + "".join("%22.1f" % (prev[c] / waves) for c in ctrs))
PY
cat "$REPO/gpurun_out/phase_counters.txt"
