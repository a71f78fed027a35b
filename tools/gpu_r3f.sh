#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'], 'verified', d['verified'])" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
timeout 1500 python -m pytest tests -m gpu -x -q -k "geo" 2>&1 | tail -4
echo -n "lanes4 cap64 rollout: "; run $B --lane-num 4 --capacity 64 --rate 1200 --steps 300
echo -n "lanes4 cap64 step: "; run $B --lane-num 4 --capacity 64 --rate 1200 --steps 300 --mode step
echo -n "lanes8 rollout p2 c25: "; run $B --lane-num 8 --steps 300
echo -n "lanes8 step p3: "; run $B --lane-num 8 --steps 300 --pipeline 3 --mode step
