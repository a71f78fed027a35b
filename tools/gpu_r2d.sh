#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-copy-peak --mode rollout --steps 1000 --warmup 0"
run() { "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'kern %.4f' % d['roofline']['kernel_ms'], 'alive %.1f' % d['mean_alive_per_env'])"; }
for w in 4 5; do for p in 2 3 4; do for c in 5 10 25 50; do
  echo -n "wpe $w pipeline $p chunk $c: "
  if [ $w = 5 ]; then PVE_ROLLOUT_WPE5=1 run $B --pipeline $p --chunk $c; else run $B --pipeline $p --chunk $c; fi
done; done; done 2>&1 | tee $O/sweep.txt
echo -n "driver-like K=20 W=5 wpe5 p2 chunk 5: "; PVE_ROLLOUT_WPE5=1 run python bench.py --no-cpu-baseline --no-copy-peak --mode rollout --steps 20 --warmup 5 --pipeline 2 --chunk 5 | tee -a $O/sweep.txt
echo -n "driver-like K=20 W=5 step: "; run python bench.py --no-cpu-baseline --no-copy-peak --mode step --steps 20 --warmup 5 | tee -a $O/sweep.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
