#!/bin/bash
# round 3, job e: 2 x 640 threads at an 80-VGPR budget (any SIMD placement of the two workgroups fits: 6 waves allowed, 5 resident)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" w6; do
  if [ -n "$lib" ]; then export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; else unset PTAMD_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 schedule=0 || exit 1
done
echo done
