#!/bin/bash
# round 3, job d: a fifth wave per SIMD for the kernel with the whole tree in LDS (2 x 640 threads, 96 VGPRs) now that
# the double-precision constants no longer spill; where the reference's own frame loop spends its time
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for lib in "" w5; do
  if [ -n "$lib" ]; then export PTAMD_LIB=$PWD/opencl_path_tracer_amd/libptamd_$lib.so; else unset PTAMD_LIB; fi
  echo "== lib ${lib:-default}"
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=256 reps=2 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 schedule=0 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 suspend_lanes=16 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 suspend_lanes=32 || exit 1
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 W=256 H=256 bounces=4 || exit 1
done
echo "== w5 library: parity tests"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_closed_form.py -x -q -k "not mesh1m and not config5" 2>&1 | tail -3
unset PTAMD_LIB
echo "== the reference's frame loop (generate_rays + trace_rays per sample)"
timeout -k 10 300 python3 tools/split_rate.py || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/r3d_split_trace -o t --output-format csv -- python3 tools/split_rate.py > gpurun_out/r3d_split_trace.log 2>&1 || { echo "trace failed"; exit 1; }
python3 tools/trace_summary.py gpurun_out/r3d_split_trace
echo done
