#!/bin/bash
# same-box A/B against an older source tree exported to ab_old/csrc (git show <rev>:.../csrc/<file>; not committed):
# times AB_SHAPES with the old library, then with the current one.   AB_SHAPES="|--mode step|--lane-num 8"
set -u
export TMPDIR=/tmp
PK=pve-mcc_for_unsignalized_intersection_amd
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
IFS='|' read -ra SHAPES <<< "${AB_SHAPES:-}"
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
make -s -C ab_old/csrc 2>&1 | grep -E "error"
cp $PK/libpveenv.so /tmp/new.so
for which in old new old new; do
  if [ $which = old ]; then cp ab_old/libpveenv.so $PK/libpveenv.so; else cp /tmp/new.so $PK/libpveenv.so; fi
  for args in "${SHAPES[@]}"; do
    for rep in 1 2 3; do $B $args 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%.2f' % (d['ms_per_step']*1e3), end=' ')"; done; echo " <- [$which] $args"
  done
done
