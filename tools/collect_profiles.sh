#!/bin/bash
# Run ON the MI355X box (through gpurun) to collect every measurement that profiles/ summarises.
# Usage: gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r5'   then   python tools/make_profile_summaries.py gpurun_out/r5 r05
# (the default bench command = BASELINE.md 3's id-indexed tape: k_rollout<128, 4, false, false, false, true, true>)
set -u
R=${1:-r4}
export TMPDIR=/tmp
O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
NB="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
run() { out=$1; shift; timeout 400 "$@" 2>/dev/null | tail -1 > $O/$out; }
# ---- bench lines (default = the persistent launch for 12 lanes x 128 slots; --persistent 0 = two sub-batches, one launch per chunk)
run bench_default.json $B
run bench_driver_like.json $B --steps 20 --warmup 5
run bench_chunked.json $NB --persistent 0
run bench_chunked_driver_like.json $NB --persistent 0 --steps 20 --warmup 5
run bench_step.json $B --mode step
run bench_single_launch.json $NB --mode step --pipeline 1
run bench_rollout_one_launch.json $NB --mode rollout --pipeline 1 --chunk 0
run bench_cap64.json $B --capacity 64
run bench_cap64_driver_like.json $NB --capacity 64 --steps 20 --warmup 5
run bench_cap64_step.json $NB --capacity 64 --mode step
run bench_actor.json $NB --actor
run bench_actor_chunked.json $NB --actor --persistent 0
run bench_actor_f64.json $NB --actor --obs-f64
run bench_actor_step.json $NB --actor --mode step
run bench_actor_driver_like.json $NB --actor --steps 20 --warmup 5
run bench_actor_lanes4.json $NB --actor --lane-num 4
run bench_actor_lanes4_step.json $NB --actor --lane-num 4 --mode step
run bench_actor_lanes4_chunked.json $NB --actor --lane-num 4 --persistent 0
run bench_actor_lanes8.json $NB --actor --lane-num 8
run bench_actor_lanes8_step.json $NB --actor --lane-num 8 --mode step
run bench_actor_lanes4_cap64.json $NB --actor --lane-num 4 --capacity 64 --rate 1000
run bench_trajectory.json $NB --trajectory 1
run bench_pool.json $NB --tape pool
run bench_pool_driver_like.json $NB --tape pool --steps 20 --warmup 5
run bench_lanes8.json $B --lane-num 8 --steps 300
run bench_lanes8_step.json $NB --lane-num 8 --steps 300 --pipeline 3 --mode step
run bench_lanes4.json $B --lane-num 4 --capacity 64 --rate 1200 --steps 300
run bench_lanes4_cap128.json $NB --lane-num 4 --steps 300
run bench_lanes4_step.json $NB --lane-num 4 --capacity 64 --rate 1200 --steps 300 --mode step
# (second pass after `make_profile_summaries.py` has written the counter profiles of this very build: ONLY_BENCH=1 re-runs just
#  the bench lines, which then carry roofline.traffic / roofline.binding)
[ "${ONLY_BENCH:-0}" = "1" ] && { ls $O; exit 0; }
# ---- per-kernel durations (rocprofv3 --kernel-trace --stats), same commands
st() { d=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats -d $O/$d -o r -- $NB "$@" > /dev/null 2>&1; }
st stats
st stats_driver_like --steps 20 --warmup 5
st stats_chunked --persistent 0
st stats_step --mode step
st stats_cap64 --capacity 64
st stats_actor --actor --steps 300
st stats_actor_step --actor --mode step --steps 300
st stats_actor_lanes4 --actor --lane-num 4 --steps 300
st stats_lanes8 --lane-num 8 --steps 300
st stats_lanes4 --lane-num 4 --capacity 64 --rate 1200 --steps 300
# ---- HBM counters in their own passes (kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass).  The timed launch of a
# persistent run is the LAST launch of its kernel (100 / 20 ticks); the bench line of the same command rides along for the shape
pmc() { m=$1; shift
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$m -o r -- $NB "$@" 2>/dev/null | tail -1 > $O/bench_pmc_$m.json
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$m -o r -- $NB "$@" > /dev/null 2>&1; }
pmc persist --steps 100
pmc persist_short --steps 20 --warmup 5
pmc rollout --persistent 0 --steps 100
pmc step --mode step --steps 100
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/probe -o r -- python tools/traffic_probe.py > $O/probe.log 2>&1
# ---- SQ counters (instruction mix, LDS conflicts, wait share): what binds the kernels (profiles/rNN_binding.json)
PMC_BENCH_ARGS="" bash tools/pmc_sq.sh persist > /dev/null 2>&1
PMC_BENCH_ARGS="--steps 20 --warmup 5" bash tools/pmc_sq.sh persist_short > /dev/null 2>&1
PMC_BENCH_ARGS="--persistent 0" bash tools/pmc_sq.sh rollout > /dev/null 2>&1
PMC_BENCH_ARGS="--mode step" bash tools/pmc_sq.sh step > /dev/null 2>&1
PMC_BENCH_ARGS="--actor" bash tools/pmc_sq.sh actor > /dev/null 2>&1
cp gpurun_out/pmc_sq_persist.txt gpurun_out/pmc_sq_persist_short.txt gpurun_out/pmc_sq_rollout.txt gpurun_out/pmc_sq_step.txt gpurun_out/pmc_sq_actor.txt $O/ 2>/dev/null
# ---- per-phase counters of the 12-lane tick (truncated launches), phase profiles, launch-shape A/B
bash tools/phase_counters.sh > /dev/null 2>&1; cp gpurun_out/phase_counters.txt $O/ 2>/dev/null
python tools/phase_profile.py --ticks 100 > $O/phase_profile_step.txt 2>&1
python tools/phase_profile.py --ticks 100 --many > $O/phase_profile_rollout.txt 2>&1
python tools/phase_profile.py --ticks 100 --many --capacity 64 > $O/phase_profile_rollout_cap64.txt 2>&1
python tools/phase_profile.py --ticks 100 --lane-num 8 > $O/phase_profile_lanes8.txt 2>&1
python tools/phase_profile.py --ticks 100 --lane-num 4 --capacity 64 --rate 1200 > $O/phase_profile_lanes4.txt 2>&1
# (the item-schedule variants need the knob build: PVE_TAPER_TAIL; the product library ignores the variable)
make -s -C pve-mcc_for_unsignalized_intersection_amd/csrc knobs > /dev/null 2>&1 && {
  PVE_LIBRARY_PATH=$(pwd)/build/libpveenv_knobs.so AB_SHAPES="12:6,3;12:5,3;13:5,2;9:3;6:3;17:3;14:3,3;10:;5:;12:5,2,1" python tools/ab_launch_shapes.py 2>&1 | grep -v amdgpu.ids > $O/ab_launch_shapes.txt
  PVE_LIBRARY_PATH=$(pwd)/build/libpveenv_knobs.so AB_CAP=64 AB_SHAPES="12:6,3;17:3;19:;15:5;10:6,4" python tools/ab_launch_shapes.py 2>&1 | grep -v amdgpu.ids > $O/ab_launch_shapes_cap64.txt; }
# ---- per-item timeline of one persistent call (diagnostics build of the library: -DPVE_QUEUE_TRACE)
make -s -C pve-mcc_for_unsignalized_intersection_amd/csrc trace > /dev/null 2>&1 && \
  PVE_LIBRARY_PATH=$(pwd)/build/libpveenv_trace.so python tools/persistent_trace.py 2>&1 | grep -v amdgpu.ids > $O/persistent_trace.txt
# ---- address unit / vector L1 counters of the headline command (the 16-byte pieces of the observation rows).  LAST, and opt-in
# (TA=1): a TA_* group has aborted rocprofv3 and hung in finalisation on this pool before (tools/pmc_mem.sh) -- five passes under
# `timeout 200` each must not starve the artefacts above
[ "${TA:-0}" = "1" ] && { bash tools/pmc_ta.sh persist > /dev/null 2>&1; cp gpurun_out/pmc_ta_persist.txt $O/ 2>/dev/null; }
# keep what comes back small (gpurun merges <= 64 MiB): only the rocpd databases of the raw rocprofv3 directories are needed
find $O gpurun_out/pmc_sq_* gpurun_out/phase_counters -type f \( -name "*.csv" -o -name "*.json" -o -name "*.log" -o -name "*.txt" \) -path "*_p[0-9]*" -size +200k -delete 2>/dev/null
rm -rf gpurun_out/pmc_sq_*/p*/ gpurun_out/pmc_ta_*/ gpurun_out/phase_counters/s* 2>/dev/null
du -sh gpurun_out $O 2>/dev/null
ls $O
