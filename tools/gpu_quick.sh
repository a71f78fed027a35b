#!/bin/bash
# quick GPU check of a kernel change: headline timings + the parity suite
set -u
export TMPDIR=/tmp
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak"
echo -n "rollout p2: "; run $B --mode rollout
echo -n "step p2: "; run $B --mode step
echo -n "step K20: "; run $B --mode step --steps 20 --warmup 5
echo -n "cap64 rollout p1: "; run $B --capacity 64 --mode rollout --pipeline 1
if [ "${1:-}" != "notest" ]; then
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
fi
