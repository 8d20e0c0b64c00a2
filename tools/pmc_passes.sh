#!/bin/bash
# usage: tools/pmc_passes.sh <tag> <prof_render args...>   -- one rocprofv3 --pmc pass per counter set
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ has 8 slots).  Output: gpurun_out/pmc_<tag>_<set>/
set -e
tag=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_${tag}_$i -o p --output-format csv -- python3 tools/prof_render.py "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  tail -1 gpurun_out/pmc_${tag}_$i.log
done
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_* > gpurun_out/pmc_${tag}_summary.txt
cat gpurun_out/pmc_${tag}_summary.txt
