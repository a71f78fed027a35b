#!/bin/bash
# The whole profile collection in ONE gpurun call, returning only text (the raw rocprofv3 databases exceed what gpurun merges back):
#   gpurun --timeout 3000 -- 'bash tools/collect_all.sh r5 r05'      then      cp gpurun_out/r5_profiles/* profiles/
# 1. tools/collect_profiles.sh (bench lines, kernel stats, counter passes, phase profiles, launch-shape A/B, queue trace)
# 2. tools/make_profile_summaries.py on the box: profiles/<prefix>_* incl. traffic.json / binding.json of THIS build
# 3. the bench lines again (ONLY_BENCH=1): they now carry roofline.traffic / roofline.binding from step 2's fingerprint-matched files
set -u
R=${1:-r5}; P=${2:-r05}
bash tools/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
python tools/make_profile_summaries.py gpurun_out/$R $P > gpurun_out/summaries_$R.log 2>&1
ONLY_BENCH=1 bash tools/collect_profiles.sh $R >> gpurun_out/collect_$R.log 2>&1
mkdir -p gpurun_out/${R}_profiles
for f in gpurun_out/$R/bench_*.json; do cp $f gpurun_out/${R}_profiles/${P}_$(basename $f); done
cp profiles/${P}_* gpurun_out/${R}_profiles/ 2>/dev/null
for f in gpurun_out/$R/bench_*.json; do cp $f gpurun_out/${R}_profiles/${P}_$(basename $f); done      # (the second pass wins)
find gpurun_out -name "*.db" -delete 2>/dev/null
rm -rf gpurun_out/$R/stats* gpurun_out/$R/fetch_* gpurun_out/$R/write_* gpurun_out/$R/probe 2>/dev/null
du -sh gpurun_out 2>/dev/null
tail -3 gpurun_out/summaries_$R.log
