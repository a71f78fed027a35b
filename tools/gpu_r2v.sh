#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "actor" 2>&1 | grep -v "^$" | tail -12
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['mode'], 'ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'], 'ovf', d['overflow'])"; }
B="python bench.py --no-cpu-baseline --no-copy-peak --actor --steps 300"
echo -n "actor p1: "; run $B --pipeline 1
echo -n "actor p2: "; run $B --pipeline 2
echo -n "actor p3: "; run $B --pipeline 3
echo -n "actor p4: "; run $B --pipeline 4
