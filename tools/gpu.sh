#!/bin/bash
# The one GPU-box helper (run through gpurun): parity suite, bench shapes, same-box A/B of build variants.
#   bash tools/gpu.sh test [pytest args]                 the -m gpu suite
#   bash tools/gpu.sh shapes "<args>|<args>|..." [reps]  every bench shape `reps` times (default 3): us per tick, exit code, verified
#   bash tools/gpu.sh ab "<EXTRA>|<EXTRA>|..." "<args>|<args>|..." [reps]
#                                                        rebuild libpveenv.so with each set of extra compiler flags (the default
#                                                        build is always the last variant) and time the shapes on THIS box
# '' (empty) is the default shape / the default build.  Timings are printed with bench.py's exit code and its `verified`
# field: a run that failed or did not verify shows as such instead of as a number.
set -u
export TMPDIR=/tmp
CSRC=pve-mcc_for_unsignalized_intersection_amd/csrc
B="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"

time_shape() {   # $1 = bench args, $2 = reps
  local args="$1" reps="${2:-3}" rep out rc
  for rep in $(seq 1 "$reps"); do
    $B $args > /tmp/gpu_sh_out.txt 2>/tmp/gpu_sh_err.txt; rc=$?      # (bench.py's own exit code, not the tail's)
    out=$(tail -1 /tmp/gpu_sh_out.txt)
    python - "$rc" "$out" <<'PY'
import json, sys
rc, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print("%.2f us (rc %s, verified %s, ovf %s)" % (d["ms_per_step"] * 1e3, rc, d.get("verified"), d.get("overflow")), end="  ")
except Exception:
    print("FAILED (rc %s): %s" % (rc, line[:120]), end="  ")
PY
  done
  echo " <- $args"
}

case "${1:-}" in
  test)
    shift
    timeout 2400 python -m pytest tests -m gpu -x -q "$@" 2>&1 | tail -6
    ;;
  shapes)
    IFS='|' read -ra SHAPES <<< "${2:-}"
    [ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
    for a in "${SHAPES[@]}"; do time_shape "$a" "${3:-3}"; done
    ;;
  ab)
    IFS='|' read -ra VARS <<< "${2:-}"
    IFS='|' read -ra SHAPES <<< "${3:-}"
    VARS+=("")
    [ ${#SHAPES[@]} -eq 0 ] && SHAPES=("")
    for v in "${VARS[@]}"; do
      touch $CSRC/pve_hip.hip
      if ! make -s -C $CSRC EXTRA="$v" > /tmp/gpu_sh_make.txt 2>&1; then
        echo "BUILD FAILED for [$v]:"; grep -m5 -E "error" /tmp/gpu_sh_make.txt; continue
      fi
      echo "== build [$v]"
      for a in "${SHAPES[@]}"; do time_shape "$a" "${4:-3}"; done
    done
    ;;
  *)
    sed -n 2,10p "$0"; exit 2
    ;;
esac
