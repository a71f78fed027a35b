#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
python tools/phase_profile.py --ticks 100 --many > $O/phase_many.txt 2>&1
python tools/phase_profile.py --ticks 100 > $O/phase_step.txt 2>&1
python tools/phase_profile.py --ticks 100 --many --envs 2048 > $O/phase_many_2048.txt 2>&1
python tools/phase_profile.py --ticks 100 --many --capacity 64 > $O/phase_many_c64.txt 2>&1
cat $O/phase_many.txt $O/phase_step.txt $O/phase_many_2048.txt | grep -v "amdgpu.ids"
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
