#!/bin/bash
# Compile ONE variant of k_rollout (or, GEO=1, of k_rollout_geo) and print its resource usage (VGPRs, spills, scratch, LDS, occupancy);
# ISA in /tmp/probe/.
#   bash tools/probe_kernel.sh "128, 5, false, false, false, true, true" [extra hipcc flags]
#   GEO=1 bash tools/probe_kernel.sh "128, true, 4, false, false, true, true"
set -e
V="${1:-128, 5, false, false, false, true, true}"; shift || true
CSRC=/root/repo/pve-mcc_for_unsignalized_intersection_amd/csrc
mkdir -p /tmp/probe && cd /tmp/probe
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off -fno-strict-aliasing -mllvm -disable-machine-licm --offload-arch=gfx950 \
  --cuda-device-only -c -save-temps -Rpass-analysis=kernel-resource-usage "-DPVE_PROBE_$( [ -n "${GEO:-}" ] && echo GEO || echo ONE )=$V" "$@" \
  "$(realpath $CSRC)/pve_hip.hip" -o /tmp/probe/one.o 2>&1 | grep -E "remark|error" | sed -e 's/.*remark: //' | grep -E "Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size|error" 
