#!/bin/bash
# Run ON the MI355X box (through gpurun) to collect every measurement that profiles/ summarises.
# Usage: gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r1'
set -u
R=${1:-r1}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
timeout 400 python bench.py 2>&1 | tail -1 > $O/bench_default.json
timeout 400 python bench.py --pipeline 1 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_single_launch.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_single -o r -- python bench.py --pipeline 1 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- python bench.py --no-cpu-baseline > $O/stats.log 2>&1
# PMC counters in their own passes (kernel-trace only), FETCH_SIZE and WRITE_SIZE cannot share a pass
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o r -- python bench.py --steps 40 --warmup 300 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o r -- python bench.py --steps 40 --warmup 300 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/probe -o r -- python tools/traffic_probe.py > $O/probe.log 2>&1
timeout 300 python bench.py --capacity 64 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cap64.json
timeout 300 python bench.py --actor --no-cpu-baseline 2>&1 | tail -1 > $O/bench_actor.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_actor -o r -- python bench.py --actor --no-cpu-baseline --steps 300 > /dev/null 2>&1
python tools/phase_profile.py --ticks 50 > $O/phase_profile.txt 2>&1
# general-geometry kernel (4 / 8 lanes)
timeout 300 python bench.py --lane-num 8 --steps 300 2>&1 | tail -1 > $O/bench_lanes8.json
timeout 300 python bench.py --lane-num 4 --capacity 64 --steps 300 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_lanes4.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_lanes8 -o r -- python bench.py --lane-num 8 --steps 300 --no-cpu-baseline > /dev/null 2>&1
ls $O
