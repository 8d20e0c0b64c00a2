#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
echo "== form2"; PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_form2.so timeout -k 10 300 python tools/sweep_suspend.py cb 0 2>&1 | grep -v "lds_scene\|balance\|nodes/seg"
echo "== form1"; timeout -k 10 300 python tools/sweep_suspend.py cb 0,24 2>&1 | grep -v "lds_scene\|balance\|nodes/seg"
echo "== var3"; PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_var3.so timeout -k 10 300 python tools/sweep_suspend.py cb 0,24 2>&1 | grep -v "lds_scene\|balance\|nodes/seg"
export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_var3.so
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d gpurun_out/pmc_var3_1 -o p --output-format csv -- python3 tools/prof_render.py spp=64 reps=2 suspend_lanes=0 > gpurun_out/pmc_var3_1.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_var3_1
