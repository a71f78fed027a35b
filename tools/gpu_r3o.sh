#!/bin/bash
# round 3: steady-state launch shapes of the headline (sub-batches x ticks per launch)
set -u
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
for p in 2 3 4; do for c in 15 25 50; do
  for rep in 1 2 3; do $B --pipeline $p --chunk $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- pipeline $p chunk $c"
done; done
