"""Suspension threshold x pass length on the Cornell box after the round's last kernel changes (development tool)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from opencl_path_tracer_amd import scenes
from sweep import run
cb = scenes.cornell_box()
for k in (8, 12, 16, 20, 24, 32):
    run(1920, 1080, 8, 64, cb, reps=3, suspend_lanes=k)
for c in (16, 32, 64):
    run(1920, 1080, 8, 64, cb, reps=3, chunk_spp=c)
run(1920, 1080, 8, 256, cb, reps=1)
run(1920, 1080, 8, 64, cb, reps=3, flat_list=0)
