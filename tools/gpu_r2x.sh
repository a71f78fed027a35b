#!/bin/bash
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
for g in 256 512 768 1024; do for p in 1 3; do echo -n "grid $g p$p: "; PVE_ACTOR_GRID=$g run $B --pipeline $p; done; done
echo -n "rollout(C loop) p3: "; run $B --pipeline 3 --mode rollout
echo -n "f64 rows p3: "; run $B --pipeline 3 --obs-f64
