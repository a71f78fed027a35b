#!/bin/bash
# round 3: A/B of a change to the geometry kernels: their parity tests, then 8 lanes and 4 lanes x 64 (rollout + step, 3 repeats)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "geo or lane" 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
for args in "--lane-num 8" "--lane-num 8 --mode step" "--lane-num 4 --capacity 64" "--lane-num 4 --capacity 64 --mode step"; do
  for rep in 1 2 3; do $B $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- $args"
done
