#!/bin/bash
# round 3: the id-indexed tape (PVE_SRC_TABLE) against the slot-indexed pool
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ctl %.1f' % d['mean_ctl_per_env'], 'ovf', d['overflow'], 'verified', d['verified'])" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
timeout 1500 python -m pytest tests -m gpu -x -q -k "step_many_equals_single_ticks" 2>&1 | tail -4
echo -n "id-sin: "; run $B
echo -n "pool: "; run $B --tape pool
echo -n "id-sin driver: "; run $B --steps 20 --warmup 5
echo -n "pool driver: "; run $B --tape pool --steps 20 --warmup 5
echo -n "id-sin cap64: "; run $B --capacity 64
echo -n "id-sin traj: "; run $B --trajectory 1
