#!/bin/bash
# the driver's short protocol (--steps 20 --warmup 5): which launch shape serves it best
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --steps 20 --warmup 5"
echo -n "step p2: "; run $B --mode step
echo -n "step p2: "; run $B --mode step
for p in 2 3 4; do for c in 2 4 5 10; do echo -n "rollout p$p c$c: "; run $B --mode rollout --pipeline $p --chunk $c; done; done
