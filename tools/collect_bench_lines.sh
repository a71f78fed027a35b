#!/bin/bash
# Second pass of a profile collection: only the bench lines, run AFTER profiles/rNN_traffic.json and rNN_binding.json of the
# same build were committed, so that bench.py attaches the counter traffic and the binding roofline (fingerprint match).
# Usage: gpurun --timeout 1500 -- 'bash tools/collect_bench_lines.sh r3'
set -u
R=${1:-r3}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
NB="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
timeout 400 $B 2>/dev/null | tail -1 > $O/bench_default.json
timeout 400 $B --mode step 2>/dev/null | tail -1 > $O/bench_step.json
timeout 400 $B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_like.json
timeout 300 $NB --actor 2>/dev/null | tail -1 > $O/bench_actor.json
timeout 300 $NB --actor --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_actor_driver_like.json
ls $O
