#!/bin/bash
# Same-box A/B of two builds of the library, alternating: bash tools/ab_libs.sh "<lib A>|<lib B>" "<bench args>|<bench args>" [reps]
# (PRODUCT = the product library next to the package).  Prints us per tick, bench.py's `verified` and `overflow` per run.
export TMPDIR=/tmp
IFS='|' read -ra LIBS <<< "$1"; IFS='|' read -ra SHAPES <<< "$2"
[ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
us() { tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f us verified=%s ovf=%s" % (d["ms_per_step"]*1e3, d.get("verified"), d.get("overflow")))'; }
for a in "${SHAPES[@]}"; do for rep in $(seq 1 ${3:-3}); do for l in "${LIBS[@]}"; do
  if [ "$l" != "PRODUCT" ]; then echo "[$a] $(basename $l): $(PVE_LIBRARY_PATH=$PWD/$l $B $a 2>/dev/null | us)"; else echo "[$a] product: $($B $a 2>/dev/null | us)"; fi
done; done; done
