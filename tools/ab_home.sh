#!/bin/bash
# Same-box A/B of the HOME build of the headline kernel (k_rollout<128, 5, ..>: 93 VGPR, LDS homes, 10 workgroups / CU) against
# the 8-workgroup kernel: knob build, PVE_ROLLOUT_WPE5 = 0 / 1, alternating.  Usage (through gpurun): bash tools/ab_home.sh [reps]
export TMPDIR=/tmp PVE_LIBRARY_PATH=$PWD/build/libpveenv_knobs.so
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
us() { tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f us verified=%s ovf=%s" % (d["ms_per_step"]*1e3, d.get("verified"), d.get("overflow")))'; }
for shape in "--steps 20 --warmup 5" "--steps 1000 --warmup 300"; do
  for rep in $(seq 1 ${1:-3}); do
    for w in 0 1; do echo "WPE5=$w [$shape] $(PVE_ROLLOUT_WPE5=$w $B $shape 2>/dev/null | us)"; done
  done
done
