#!/bin/bash
# round 3, job i: the reference's frame loop (generate_rays + trace_rays per sample) with the tile-stream schedule against lockstep
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "split or dropin or schedules or register_budgets or view or seed" 2>&1 | tail -3
python3 - <<'PY'
import sys, time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
W, H, B = 1920, 1080, 8
for name, spec in (("cornell", scenes.cornell_box()), ("mesh100k", scenes.displaced_grid_mesh(100000))):
    for opts in ({"schedule": 0}, {}, {"lds_block": 512}, {"suspend_lanes": 16}, {"suspend_lanes": 32}):
        sc = api.Scene(W, H).load(spec)
        for k, v in opts.items():
            sc.set_option(k, v)
        sc.iterations = B
        sc.render(4, fused=False); sc.sync()
        t = time.time(); sc.render(32, fused=False); sc.sync(); dt = time.time() - t
        print("%s %-22s: 32 x (generate_rays + trace_rays): %.1f Msamples/s (%.3f ms per sample)" % (name, opts, W * H * 32 / dt / 1e6, dt / 32 * 1e3), flush=True)
    sc = api.Scene(W, H).load(spec)
    sc.iterations = B
    sc.render(32); sc.sync()
    t = time.time(); sc.render(64); sc.sync(); dt = time.time() - t
    print("%s: render(64): %.1f Msamples/s" % (name, W * H * 64 / dt / 1e6), flush=True)
PY
echo done
