"""Wavefront variant: chains of the frame on separate HIP streams (option wf_streams), wall-clock rate at 1920x1080.
usage: python tools/sweep_wf_streams.py <scene,...> [--ws 1,2,4,8] [--reps 3]     scene: cornell mesh100k mesh1m
(more than four streams only pay with GPU_MAX_HW_QUEUES raised above the runtime's default of 4 in the environment)"""
import argparse
import os
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes")
ap.add_argument("--ws", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
CASES = {"cornell": (scenes.cornell_box, 8, 8), "mesh100k": (lambda: scenes.displaced_grid_mesh(100000), 8, 4),
         "mesh1m": (lambda: scenes.displaced_grid_mesh(1000000), 16, 2)}
for name in args.scenes.split(","):
    make, bounces, spp = CASES[name]
    sc = api.Scene(1920, 1080).load(make())
    sc.iterations = bounces
    sc.set_option("variant", 1)
    for ws in [int(x) for x in args.ws.split(",")]:
        sc.set_option("wf_streams", ws)
        sc.render(spp)
        sc.sync()
        t = time.time()
        for _ in range(args.reps):
            sc.render(spp)
        enq = time.time() - t           # the host's share: every launch of the passes enqueued
        sc.sync()
        dt = time.time() - t
        print("%-9s wf_streams %d (GPU_MAX_HW_QUEUES=%s): %8.1f Msamples/s   (enqueue %.2f ms of %.2f ms per sample pass)"
              % (name, ws, os.environ.get("GPU_MAX_HW_QUEUES", "default"), 1920 * 1080 * spp * args.reps / dt / 1e6, enq * 1e3 / (spp * args.reps), dt * 1e3 / (spp * args.reps)), flush=True)
    sc.close()
