#!/bin/bash
# every bench configuration of DESIGN.md 5, one line each (no CPU baseline, no copy-peak measurement)
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "rollout p2 c25: "; run $B --mode rollout
echo -n "step p2: "; run $B --mode step
echo -n "step p3: "; run $B --mode step --pipeline 3
echo -n "step p1: "; run $B --mode step --pipeline 1
echo -n "step K20: "; run $B --mode step --steps 20 --warmup 5
echo -n "cap64 rollout p1: "; run $B --capacity 64 --mode rollout --pipeline 1
echo -n "cap64 step p2: "; run $B --capacity 64 --mode step
echo -n "lanes8 p3: "; run $B --lane-num 8 --pipeline 3 --steps 300
echo -n "lanes4 cap64 p2: "; run $B --lane-num 4 --capacity 64 --steps 300
echo -n "actor p2: "; run $B --actor --pipeline 2 --steps 300
echo -n "actor p3: "; run $B --actor --pipeline 3 --steps 300
