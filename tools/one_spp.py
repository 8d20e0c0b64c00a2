import sys, time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
W, H, B = 1920, 1080, 8
spec = scenes.cornell_box()
for opts in ({}, {"schedule": 0}, {"schedule": 1}, {"lds_block": 512}):
    sc = api.Scene(W, H).load(spec)
    for k, v in opts.items():
        sc.set_option(k, v)
    sc.iterations = B
    for _ in range(4): sc.render(1)
    sc.sync()
    t = time.time()
    for _ in range(64): sc.render(1)
    sc.sync(); dt = time.time() - t
    print("fused render(1) x 64 %-18s: %.1f Msamples/s (%.3f ms per sample)" % (opts, W * H * 64 / dt / 1e6, dt / 64 * 1e3), flush=True)
    for _ in range(4): sc.render(1, fused=False)
    sc.sync()
    t = time.time()
    sc.render(64, fused=False)
    sc.sync(); dt = time.time() - t
    print("generate_rays + trace_rays x 64 %-18s: %.1f Msamples/s (%.3f ms per sample)" % (opts, W * H * 64 / dt / 1e6, dt / 64 * 1e3), flush=True)
