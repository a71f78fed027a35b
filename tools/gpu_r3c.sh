#!/bin/bash
# round 3: headline timings + the full parity suite
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { "$@" 2>gpurun_out/err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['config']['mode'], 'tpl', d['config']['ticks_per_launch'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % r['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'], 'verified', d['verified'])" || tail -5 gpurun_out/err.log; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "rollout: "; run $B
echo -n "step: "; run $B --mode step
echo -n "driver rollout: "; run $B --steps 20 --warmup 5
echo -n "actor rollout c25: "; run $B --actor --mode rollout --chunk 25
echo -n "cap64 rollout p1: "; run $B --capacity 64 --pipeline 1
if [ "${1:-}" != "notest" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
fi
