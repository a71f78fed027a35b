#!/bin/bash
# Round 6: ONE profile collection on the final sources, trimmed to what DESIGN.md / bench.py cite.
#   gpurun --timeout 2700 -- 'bash tools/collect_r06.sh'     ->  gpurun_out/r6_profiles/r06_*  (then: cp gpurun_out/r6_profiles/* profiles/)
# 1. GPU test suite  2. bench lines  3. rocprofv3 --kernel-trace --stats of the default and the driver's command
# 4. HBM counters (FETCH_SIZE / WRITE_SIZE in passes of their own) + calibration probe  5. SQ counters (what binds)
# 6. summaries + traffic.json / binding.json of THIS build  7. the headline bench lines again (they now carry the counter figures)
set -u
export TMPDIR=/tmp
R=r6; P=r06; O=gpurun_out/$R
mkdir -p $O
B="python bench.py"
NB="python bench.py --no-cpu-baseline --no-copy-peak --no-companion"
run() { out=$1; shift; timeout 400 "$@" 2>/dev/null | tail -1 > $O/$out; }
bash tools/gpu.sh test > $O/gpu_suite.txt 2>&1
lines() {
  run bench_default.json $B
  run bench_driver_like.json $B --steps 20 --warmup 5
}
lines
run bench_step.json $NB --mode step
run bench_chunked.json $NB --persistent 0
run bench_pool.json $NB --tape pool
run bench_cap64.json $NB --capacity 64
run bench_cap64_driver_like.json $NB --capacity 64 --steps 20 --warmup 5
run bench_actor.json $NB --actor
run bench_actor_driver_like.json $NB --actor --steps 20 --warmup 5
run bench_lanes8.json $NB --lane-num 8 --steps 300
run bench_lanes4.json $NB --lane-num 4 --capacity 64 --rate 1200 --steps 300
run bench_lanes4_cap128.json $NB --lane-num 4 --steps 300
run bench_actor_lanes4.json $NB --actor --lane-num 4
run bench_actor_lanes8.json $NB --actor --lane-num 8
st() { d=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats -d $O/$d -o r -- $NB "$@" > /dev/null 2>&1; }
st stats
st stats_driver_like --steps 20 --warmup 5
st stats_cap64 --capacity 64 --steps 20 --warmup 5
st stats_actor --actor --steps 300
pmc() { m=$1; shift
  timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_$m -o r -- $NB "$@" 2>/dev/null | tail -1 > $O/bench_pmc_$m.json
  timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write_$m -o r -- $NB "$@" > /dev/null 2>&1; }
pmc persist --steps 100
pmc persist_short --steps 20 --warmup 5
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/probe -o r -- python tools/traffic_probe.py > $O/probe.log 2>&1
PMC_BENCH_ARGS="" bash tools/pmc_sq.sh persist > /dev/null 2>&1
PMC_BENCH_ARGS="--steps 20 --warmup 5" bash tools/pmc_sq.sh persist_short > /dev/null 2>&1
PMC_BENCH_ARGS="--actor" bash tools/pmc_sq.sh actor > /dev/null 2>&1
cp gpurun_out/pmc_sq_persist.txt gpurun_out/pmc_sq_persist_short.txt gpurun_out/pmc_sq_actor.txt $O/ 2>/dev/null
python tools/phase_profile.py --ticks 100 --many > $O/phase_profile_rollout.txt 2>&1
PVE_LIBRARY_PATH=$(pwd)/build/libpveenv_trace.so TRACE_K=20 TRACE_CHUNK=12 python tools/persistent_trace.py 2>&1 | grep -v amdgpu.ids > $O/persistent_trace.txt
python tools/make_profile_summaries.py $O $P > gpurun_out/summaries_$R.log 2>&1
lines                                             # (second pass: roofline.traffic / roofline.binding from this build's counter files)
mkdir -p gpurun_out/${R}_profiles
cp profiles/${P}_* gpurun_out/${R}_profiles/ 2>/dev/null
for f in $O/bench_*.json $O/gpu_suite.txt; do cp $f gpurun_out/${R}_profiles/${P}_$(basename $f); done
find gpurun_out -name "*.db" -delete 2>/dev/null
rm -rf $O/stats* $O/fetch_* $O/write_* $O/probe gpurun_out/pmc_sq_*/ 2>/dev/null
du -sh gpurun_out 2>/dev/null; tail -3 gpurun_out/summaries_$R.log; ls gpurun_out/${R}_profiles | wc -l
