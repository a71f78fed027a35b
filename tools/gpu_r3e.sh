#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'], 'verified', d['verified'])" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "lanes8 rollout p2 c25: "; run $B --lane-num 8 --steps 300 --mode rollout
export PVE_ROLLOUT_GEO_WPE5=1
echo -n "lanes8 rollout WPE5 p2 c25: "; run $B --lane-num 8 --steps 300 --mode rollout
echo -n "lanes8 rollout WPE5 p3 c25: "; run $B --lane-num 8 --steps 300 --mode rollout --pipeline 3
echo -n "lanes8 rollout WPE5 p2 c10: "; run $B --lane-num 8 --steps 300 --mode rollout --chunk 10
