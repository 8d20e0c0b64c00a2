#!/bin/bash
# round 3, step w: re-tune of the suspension threshold and the pass length for the 768-thread Cornell instance
cd "${GRAFT_REPO_ROOT:-.}"
for s in 12 16 20 24 28 32 40; do echo "suspend_lanes $s"; timeout -k 5 120 python3 tools/prof_render.py scene=cornell spp=64 reps=4 suspend_lanes=$s 2>&1 | tail -1; done
for c in 0 8 16 32 64; do echo "chunk_spp $c"; timeout -k 5 120 python3 tools/prof_render.py scene=cornell spp=64 reps=4 chunk_spp=$c 2>&1 | tail -1; done
