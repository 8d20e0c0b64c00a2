#!/bin/bash
# round 3, job a: evidence for the scratch traffic of the global-memory kernels (per waves-per-SIMD instance) and
# per-kernel numbers of the wavefront variant, all with the round-2 kernels
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
S="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for w in 4 5 6 7; do
  tools/pmc_sets.sh r3a_m100k_w$w "$S" scene=mesh100k spp=16 reps=2 waves_per_simd=$w > gpurun_out/r3a_m100k_w$w.log 2>&1 || { echo "pmc m100k w$w failed"; tail -5 gpurun_out/r3a_m100k_w$w.log; exit 1; }
  grep -h "Msamples" gpurun_out/pmc_r3a_m100k_w${w}_1.log
done
for w in 4 7; do
  tools/pmc_sets.sh r3a_m1m_w$w "$S" scene=mesh1m spp=8 bounces=16 reps=2 waves_per_simd=$w > gpurun_out/r3a_m1m_w$w.log 2>&1 || { echo "pmc m1m w$w failed"; exit 1; }
  grep -h "Msamples" gpurun_out/pmc_r3a_m1m_w${w}_1.log
done
# wavefront variant: per-kernel time + HBM-side bytes
for sc in cornell mesh100k; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r3a_wf_${sc}_trace -o t --output-format csv -- python3 tools/prof_render.py scene=$sc spp=8 reps=2 variant=1 > gpurun_out/r3a_wf_${sc}_trace.log 2>&1 || { echo "wf trace $sc failed"; exit 1; }
  python3 tools/trace_summary.py gpurun_out/r3a_wf_${sc}_trace > gpurun_out/r3a_wf_${sc}_trace_summary.txt; cat gpurun_out/r3a_wf_${sc}_trace_summary.txt
  tools/pmc_sets.sh r3a_wf_$sc "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" scene=$sc spp=8 reps=2 variant=1 > gpurun_out/r3a_wf_$sc.log 2>&1 || { echo "wf pmc $sc failed"; exit 1; }
done
echo done
