#!/bin/bash
# round 3: quick A/B of a kernel change: the parity tests that exercise it, then the three headline shapes (3 repeats each)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q ${R3I_TESTS:-} 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
for args in "" "--mode step" "--capacity 64" ${R3I_EXTRA:-}; do
  for rep in 1 2 3; do $B $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- default $args"
done
