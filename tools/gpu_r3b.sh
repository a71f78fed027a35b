#!/bin/bash
# round 3: the actor rewrite (32-wide tile, packed weights) and the closed loop inside k_rollout
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'], 'verified', d['verified'])" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
timeout 1500 python -m pytest tests -m gpu -x -q -k "actor or step_many or closed_loop" 2>&1 | tail -6
echo -n "actor step: "; run $B --actor
echo -n "actor rollout c25: "; run $B --actor --mode rollout --chunk 25
echo -n "actor rollout c10: "; run $B --actor --mode rollout --chunk 10
echo -n "actor rollout c25 p3: "; run $B --actor --mode rollout --chunk 25 --pipeline 3
echo -n "actor rollout f64: "; run $B --actor --mode rollout --chunk 25 --obs-f64
