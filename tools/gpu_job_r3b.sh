#!/bin/bash
# round 3, job b: path state parked in global memory (default) against the register / scratch form (libptamd_nopark.so)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" nopark; do
  [ -n "$lib" ] && export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so
  echo "== lib ${lib:-default}"
  for w in 5 6 7; do
    timeout -k 10 200 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 waves_per_simd=$w || exit 1
  done
  for w in 6 7; do
    timeout -k 10 200 python3 tools/prof_render.py scene=mesh1m spp=8 bounces=16 reps=3 waves_per_simd=$w || exit 1
  done
done
unset PTAMD_LIB
S="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
for w in 6 7; do
  tools/pmc_sets.sh r3b_m100k_w$w "$S" scene=mesh100k spp=16 reps=2 waves_per_simd=$w > gpurun_out/r3b_m100k_w$w.log 2>&1 || { echo "pmc failed"; exit 1; }
done
tools/pmc_sets.sh r3b_m1m_w7 "$S" scene=mesh1m spp=8 bounces=16 reps=2 waves_per_simd=7 > gpurun_out/r3b_m1m_w7.log 2>&1 || { echo "pmc failed"; exit 1; }
echo done
