#!/bin/bash
# Round 6: what the profile collection (tools/collect_r06.sh) does not cover -- run in the same gpurun call, after it:
# the LDS-granule probe, the HOME kernel's same-box A/B, the trainer's roll-out speeds per layout / source, the randomised soak.
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_profiles
mkdir -p $O
./build/occ_probe > $O/r06_occupancy_probe.txt 2>&1
bash tools/ab_home.sh 2 > $O/r06_ab_home_vs_wpe4.txt 2>&1
{
  for LN in 12 4 8; do
    for SRC in pool actor; do
      LANE_NUM=$LN SOURCE=$SRC timeout 300 python tools/trainer_rollout_speed.py "final sources" 2>&1 | tail -1
    done
  done
} > $O/r06_trainer_rollout_final.txt
timeout 1500 python tools/soak_random.py --runs 60 --many 300 --seed 66 2>&1 | grep -v amdgpu.ids > $O/r06_soak_random.txt
tail -2 $O/r06_soak_random.txt
