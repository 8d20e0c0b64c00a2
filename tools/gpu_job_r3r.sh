#!/bin/bash
# round 3, job r: is LEAN (running mean folded into colors[] per sample ...) right for the six-wave LDS-tree instance?  (libptamd_lean7.so: LEAN from 7 waves only);
# suspension threshold of the mesh kernels re-swept
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" lean7; do
  if [ -n "$lib" ]; then export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; else unset PTAMD_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=256 reps=2 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 waves_per_simd=6 || exit 1
done
unset PTAMD_LIB
for sl in 16 24 32 40; do
  timeout -k 10 200 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 suspend_lanes=$sl || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=mesh1m spp=8 bounces=16 reps=3 suspend_lanes=$sl || exit 1
done
echo done
