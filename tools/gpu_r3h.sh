#!/bin/bash
# round 3: launch shapes for the driver's short protocol (--steps 20 --warmup 5)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify --steps 20 --warmup 5"
for p in 2 3; do for c in 3 4 5 6 7 10; do
  for rep in 1 2 3; do $B --pipeline $p --chunk $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- pipeline $p chunk $c"
done; done
