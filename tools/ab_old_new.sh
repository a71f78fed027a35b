#!/bin/bash
# same-box A/B of prebuilt libraries against the in-tree one, alternating runs (the compiler's register allocation near the
# 128-VGPR limit flips between 0 and ~10 spilled registers on source changes that do not touch the kernel's semantics, so an
# A/B built from -D knobs can compare two DIFFERENT allocations of the "old" code: compare against the library actually built
# from the old commit)
#   bash tools/ab_old_new.sh "<args>|<args>|..." [reps] [lib ...]     libs default to build/libpveenv_old.so
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion --no-verify"
IFS='|' read -ra SHAPES <<< "${1:-}"
REPS=${2:-3}
[ $# -ge 2 ] && shift 2 || shift $#          # (only the arguments that are present)
LIBS=("$@"); [ ${#LIBS[@]} -eq 0 ] && LIBS=(build/libpveenv_old.so)
us() { grep "^{" | tail -1 | python -c "import json,sys; print('%.2f' % (json.loads(sys.stdin.read())['ms_per_step'] * 1e3))"; }
for a in "${SHAPES[@]}"; do
  for rep in $(seq 1 "$REPS"); do
    for l in "${LIBS[@]}"; do echo -n "$(basename $l .so | sed s/libpveenv_//) $(PVE_LIBRARY_PATH=$PWD/$l $B $a 2>/dev/null | us) "; done
    echo -n "tree $($B $a 2>/dev/null | us) | "
  done
  echo " <- $a"
done
