#!/bin/bash
# VGPR / SGPR / scratch / LDS / occupancy of every kernel of a source file (compile-time remarks)
# usage: tools/kernel_resources.sh opencl_path_tracer_amd/csrc/pt_kernels.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sdwa-peephole=0 -Iinclude -Iopencl_path_tracer_amd/csrc \
  -Rpass-analysis=kernel-resource-usage "$@" -c -o /dev/null "$f" 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size" | \
  sed -e 's/.*remark: [^ ]* //' | paste - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | c++filt | cut -c1-260
