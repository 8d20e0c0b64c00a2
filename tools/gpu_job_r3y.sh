#!/bin/bash
# round 3, step y: selects that read a SALU-written VCC (21 clocks) replaced: SDWA peephole off (libptamd_nosdwa.so = that flag alone),
# + sel64 in the exact triangle test / the 4-wide node visit (default build)
cd "${GRAFT_REPO_ROOT:-.}"
for lib in opencl_path_tracer_amd/libptamd_nosdwa.so "" opencl_path_tracer_amd/libptamd_nosdwa.so ""; do
  echo "== lib ${lib:-default}"
  PTAMD_LIB=$lib timeout -k 5 120 python3 tools/prof_render.py scene=cornell spp=64 reps=4 2>&1 | tail -1
  PTAMD_LIB=$lib timeout -k 5 120 python3 tools/prof_render.py scene=mesh100k spp=16 reps=3 2>&1 | tail -1
  PTAMD_LIB=$lib timeout -k 5 120 python3 tools/prof_render.py scene=mesh1m spp=8 bounces=16 reps=3 2>&1 | tail -1
done
