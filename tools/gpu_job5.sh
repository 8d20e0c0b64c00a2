#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in form2 form1; do
  export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${lib}_$i -o p --output-format csv -- python3 tools/prof_render.py spp=64 reps=2 suspend_lanes=0 > gpurun_out/pmc_${lib}_$i.log 2>&1
    grep "^scene" gpurun_out/pmc_${lib}_$i.log
  done
  python3 tools/pmc_summary.py gpurun_out/pmc_${lib}_1 gpurun_out/pmc_${lib}_2 > gpurun_out/pmc_${lib}_summary.txt
  cat gpurun_out/pmc_${lib}_summary.txt
done
