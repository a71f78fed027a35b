#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-copy-peak --mode rollout --pipeline 1 --steps 300 --warmup 0"
for n in 1024 1536 1792 2048 2304 2560 3072 4096; do
  $B --envs $n 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('envs', d['config']['envs_per_gpu'], 'ms/step %.4f' % d['ms_per_step'], 'alive %.1f' % d['mean_alive_per_env'])"
done 2>&1 | tee $O/sweep.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
