#!/bin/bash
# closed loop (BASELINE config 5): sub-batch count sweep
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
for p in 1 2 3 4 6 8; do echo -n "actor p$p: "; run $B --pipeline $p; done
for g in 256 512; do echo -n "actor p4 grid $g: "; PVE_ACTOR_GRID=$g run $B --pipeline 4; done
