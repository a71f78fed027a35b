#!/bin/bash
# first GPU pass of round 2: tests + bench variants (step vs rollout)
set -u
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
B="python bench.py --no-cpu-baseline --no-copy-peak"
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 $B --mode step > $O/bench_step.json 2>&1
timeout 300 $B --mode rollout --pipeline 1 > $O/bench_roll_p1.json 2>&1
timeout 300 $B --mode rollout --pipeline 2 > $O/bench_roll_p2.json 2>&1
PVE_ROLLOUT_WPE5=1 timeout 300 $B --mode rollout --pipeline 2 > $O/bench_roll_p2_w5.json 2>&1
PVE_ROLLOUT_WPE5=1 timeout 300 $B --mode rollout --pipeline 1 > $O/bench_roll_p1_w5.json 2>&1
PVE_NO_ROLLOUT_KERNEL=1 timeout 300 $B --mode rollout --pipeline 2 > $O/bench_roll_hostloop.json 2>&1
timeout 300 $B --mode rollout --steps 20 --warmup 5 > $O/bench_roll_driver.json 2>&1
timeout 300 $B --capacity 64 --mode step > $O/bench_cap64_step.json 2>&1
timeout 300 $B --capacity 64 --mode rollout --pipeline 1 > $O/bench_cap64_roll.json 2>&1
timeout 300 $B --actor --mode step --steps 300 > $O/bench_actor_step.json 2>&1
timeout 300 $B --lane-num 8 --steps 300 > $O/bench_lanes8.json 2>&1
for f in $O/bench_*.json; do echo "== $f"; tail -1 $f | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    r = d['roofline']
    print(d['config']['mode'], 'ms/step %.4f value %.3e alive %.1f pop %s ovf %s frac %.3f kern_ms %.4f peak_meas %s' % (d['ms_per_step'], d['value'], d['mean_alive_per_env'], d['population'], d['overflow'], r['frac'], r['kernel_ms'], r.get('peak_measured')))
    if 'cpu_baseline' in d: print(d['cpu_baseline'])
except Exception as e:
    print('unparsed', e)
"; done
