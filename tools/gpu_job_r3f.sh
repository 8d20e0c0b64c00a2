#!/bin/bash
# round 3, job f: 56-byte LDS nodes (default) against 64 (n64); 2 x 640 and 2 x 768 threads at 80 VGPRs
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" n64 w6 w768; do
  if [ -n "$lib" ]; then export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; else unset PTAMD_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=256 reps=2 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 schedule=0 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 W=256 H=256 bounces=4 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=16 reps=4 variant=1 || exit 1
done
for lib in w6 w768; do
  export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so
  echo "== $lib library: parity tests"
  timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_closed_form.py -x -q -k "not mesh1m and not config5" 2>&1 | tail -3
done
echo done
