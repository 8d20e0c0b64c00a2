#!/bin/bash
# round 3, job c: lean instances + scalar-register DP constants (default) against constants left to the allocator
# (libptamd_vconst.so) and round 2's kernels (libptamd_r2k.so); the 8-wave instance; counters of the new default
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" vconst r2k; do
  if [ -n "$lib" ]; then export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; else unset PTAMD_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=3 || exit 1
  for w in 5 6 7; do
    timeout -k 10 200 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 waves_per_simd=$w || exit 1
  done
  for w in 6 7; do
    timeout -k 10 200 python3 tools/prof_render.py scene=mesh1m spp=8 bounces=16 reps=3 waves_per_simd=$w || exit 1
  done
done
unset PTAMD_LIB
echo "== default, 8 waves (wide_lds_entries 16), and 7 waves with the same stacks"
for w in 7 8; do
  timeout -k 10 200 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 waves_per_simd=$w wide_lds_entries=16 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=mesh1m spp=8 bounces=16 reps=3 waves_per_simd=$w wide_lds_entries=16 || exit 1
done
S="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
tools/pmc_sets.sh r3c_m100k_w7 "$S" scene=mesh100k spp=16 reps=2 waves_per_simd=7 > gpurun_out/r3c_m100k_w7.log 2>&1 || { echo "pmc failed"; exit 1; }
tools/pmc_sets.sh r3c_m1m_w7 "$S" scene=mesh1m spp=8 bounces=16 reps=2 waves_per_simd=7 > gpurun_out/r3c_m1m_w7.log 2>&1 || { echo "pmc failed"; exit 1; }
echo done
