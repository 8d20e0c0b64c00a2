"""After the flat list the tree holds only the objects: SAH visit price, chained-pass length."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from sweep import run  # noqa: E402
from opencl_path_tracer_amd import api, scenes  # noqa: E402
import sweep  # noqa: E402

cb = scenes.cornell_box()
orig_run = run


def run_pre(W, H, b, spp, spec, reps=2, **opts):
    pre = {k: opts.pop(k) for k in list(opts) if k in ("sah_visit_cost", "bvh_policy")}
    sc_opts = dict(opts)
    # sweep.run sets "pre" options before load only for bvh_policy / treelet / flat_list: do it by hand here
    sc = api.Scene(W, H)
    for k, v in pre.items():
        sc.set_option(k, v)
    sc.load(spec)
    for k, v in sc_opts.items():
        sc.set_option(k, v)
    sc.set_option("timing", 1)
    sc.iterations = b
    sc.render(2)
    sc.sync()
    sc.set_option("reset_stats", 1)
    for _ in range(reps):
        sc.render(spp)
    sc.sync()
    kms, samples, segs = sc.stat("kernel_ms"), sc.stat("samples"), sc.stat("segments")
    print("%dx%d b%d spp%d %-40s nodes=%d: %8.1f Msamples/s" % (W, H, b, spp, str({**pre, **sc_opts}), sc.stat("bvh_nodes"), samples / kms / 1e3), flush=True)


for vc in (5, 8, 10, 12, 15, 20, 30):
    run_pre(1920, 1080, 8, 64, cb, reps=3, sah_visit_cost=vc)
for pol in (2, 3):
    run_pre(1920, 1080, 8, 64, cb, reps=3, bvh_policy=pol)
for ch in (4, 8, 16, 32):
    run_pre(1920, 1080, 8, 64, cb, reps=3, chunk_spp=ch)
m = scenes.displaced_grid_mesh(100000)
for vc in (5, 10, 15, 20):
    run_pre(1920, 1080, 8, 16, m, reps=2, sah_visit_cost=vc)
for pol in (2, 3):
    run_pre(1920, 1080, 8, 16, m, reps=2, bvh_policy=pol)
