#!/bin/bash
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "step p2: "; run $B --mode step
echo -n "rollout p2 c25: "; run $B --mode rollout
echo -n "rollout p2 c25 wpe5: "; PVE_ROLLOUT_WPE5=1 run $B --mode rollout
echo -n "rollout p3 c25 wpe5: "; PVE_ROLLOUT_WPE5=1 run $B --mode rollout --pipeline 3
echo -n "rollout p2 c50 wpe5: "; PVE_ROLLOUT_WPE5=1 run $B --mode rollout --chunk 50
