#!/bin/bash
# Run ON the MI355X box (through gpurun) to collect every measurement that profiles/ summarises.
# Usage: gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r2'
set -u
R=${1:-r2}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
NB="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
# ---- bench lines
timeout 400 $B 2>/dev/null | tail -1 > $O/bench_default.json
timeout 400 $B --mode step 2>/dev/null | tail -1 > $O/bench_step.json
timeout 400 $B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_like.json
timeout 400 $NB --mode step --pipeline 1 2>/dev/null | tail -1 > $O/bench_single_launch.json
timeout 400 $NB --mode rollout --pipeline 1 --chunk 0 2>/dev/null | tail -1 > $O/bench_rollout_one_launch.json
timeout 300 $B --capacity 64 2>/dev/null | tail -1 > $O/bench_cap64.json
timeout 300 $NB --capacity 64 --mode step 2>/dev/null | tail -1 > $O/bench_cap64_step.json
timeout 300 $NB --actor 2>/dev/null | tail -1 > $O/bench_actor.json
timeout 300 $NB --actor --obs-f64 2>/dev/null | tail -1 > $O/bench_actor_f64.json
timeout 300 $NB --actor --mode step 2>/dev/null | tail -1 > $O/bench_actor_step.json
timeout 300 $NB --actor --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_actor_driver_like.json
timeout 400 $NB --trajectory 1 2>/dev/null | tail -1 > $O/bench_trajectory.json
timeout 400 $NB --tape id-sin 2>/dev/null | tail -1 > $O/bench_id_sin.json
timeout 400 $NB --tape id-sin --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_id_sin_driver_like.json
timeout 300 $B --lane-num 8 --steps 300 2>/dev/null | tail -1 > $O/bench_lanes8.json
timeout 300 $NB --lane-num 8 --steps 300 --pipeline 3 --mode step 2>/dev/null | tail -1 > $O/bench_lanes8_step.json
timeout 300 $B --lane-num 4 --capacity 64 --rate 1200 --steps 300 2>/dev/null | tail -1 > $O/bench_lanes4.json
timeout 300 $NB --lane-num 4 --capacity 64 --rate 1200 --steps 300 --mode step 2>/dev/null | tail -1 > $O/bench_lanes4_step.json
# ---- per-kernel durations (rocprofv3 --kernel-trace --stats), same commands
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o r -- $NB > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_step -o r -- $NB --mode step > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_driver_like -o r -- $NB --steps 20 --warmup 5 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_cap64 -o r -- $NB --capacity 64 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_actor -o r -- $NB --actor --steps 300 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_actor_step -o r -- $NB --actor --mode step --steps 300 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_lanes8 -o r -- $NB --lane-num 8 --steps 300 > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats_lanes4 -o r -- $NB --lane-num 4 --capacity 64 --rate 1200 --steps 300 > /dev/null 2>&1
# ---- HBM counters in their own passes (kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass)
for m in rollout step; do
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$m -o r -- $NB --mode $m --steps 100 > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$m -o r -- $NB --mode $m --steps 100 > /dev/null 2>&1
done
# (the driver's short protocol runs launches of 5 ticks: its own traffic figure)
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_rollout5 -o r -- $NB --mode rollout --chunk 5 --steps 100 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_rollout5 -o r -- $NB --mode rollout --chunk 5 --steps 100 > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/probe -o r -- python tools/traffic_probe.py > $O/probe.log 2>&1
# ---- SQ counters (instruction mix, LDS conflicts, wait share): what binds the kernels (profiles/rNN_binding.json)
bash tools/pmc_sq.sh rollout > /dev/null 2>&1
PMC_BENCH_ARGS="--steps 20 --warmup 5" bash tools/pmc_sq.sh rollout5 > /dev/null 2>&1
PMC_BENCH_ARGS="--mode step" bash tools/pmc_sq.sh step > /dev/null 2>&1
PMC_BENCH_ARGS="--actor" bash tools/pmc_sq.sh actor > /dev/null 2>&1
cp gpurun_out/pmc_sq_rollout.txt gpurun_out/pmc_sq_rollout5.txt gpurun_out/pmc_sq_step.txt gpurun_out/pmc_sq_actor.txt $O/ 2>/dev/null
# ---- phase profiles
python tools/phase_profile.py --ticks 100 > $O/phase_profile_step.txt 2>&1
python tools/phase_profile.py --ticks 100 --many > $O/phase_profile_rollout.txt 2>&1
python tools/phase_profile.py --ticks 100 --many --capacity 64 > $O/phase_profile_rollout_cap64.txt 2>&1
python tools/phase_profile.py --ticks 100 --lane-num 8 > $O/phase_profile_lanes8.txt 2>&1
python tools/phase_profile.py --ticks 100 --lane-num 4 --capacity 64 --rate 1200 > $O/phase_profile_lanes4.txt 2>&1
ls $O
