#!/bin/bash
# round 3, job g: GPU tests with the 768-thread LDS instance as default; scaling probe by instance / schedule / pass length; suspension threshold at 6 waves
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "not mesh1m_1080p" 2>&1 | tail -3
for sl in 12 16 24 32 40; do
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 suspend_lanes=$sl || exit 1
done
for ch in 16 32 64; do
  timeout -k 10 200 python3 tools/prof_render.py scene=cornell spp=64 reps=4 chunk_spp=$ch || exit 1
done
timeout -k 10 600 python3 tools/scaling_probe2.py || exit 1
echo done
