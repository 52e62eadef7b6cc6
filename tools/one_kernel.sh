#!/bin/bash
# Fast iteration on one instantiation of the wave kernel: compiles only that instantiation (seconds instead of 1.5 minutes) with the product's
# flags, keeps the ISA and prints its register / spill / scratch figures.   tools/one_kernel.sh "24, 8, 32, 1, 4, false, 1" [extra hipcc flags]
set -e
ARGS="${1:-24, 8, 32, 1, 4, false, 1}"; shift || true
D=/tmp/one_kernel; mkdir -p $D; cd $D
cat > k.hip <<EOT
#include "/root/repo/myosuite_mjx_amd/csrc/myo_common.h"
#include "/root/repo/myosuite_mjx_amd/csrc/myo_physics.h"
#include "/root/repo/myosuite_mjx_amd/csrc/myo_task_track.h"
#include "/root/repo/myosuite_mjx_amd/csrc/myo_kernel_wave.h"
template __global__ void step_kernel_w<$ARGS>(const DevModel*, const DevModelW*, DevBatch, const float*, int, int, long long*, const int*, const DevWalk*, int, SchedDev);
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -ffp-contract=on -fgpu-flush-denormals-to-zero -mllvm -disable-machine-licm -fno-slp-vectorize \
  --cuda-device-only -S -o k.s k.hip "$@"
grep -E "^\s+\.(sgpr|vgpr)_(count|spill_count)|private_segment_fixed_size|\.group_segment_fixed_size" k.s | tr -s ' \n' ' '; echo
echo "spill stores: $(grep -c 'Folded Spill' k.s)  reloads: $(grep -c 'Folded Reload' k.s)  lines: $(wc -l < k.s)"
