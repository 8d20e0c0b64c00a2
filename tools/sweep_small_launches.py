"""Schedule against samples per launch (development tool): render(n) x (64 / n) for small n, lockstep and suspend."""
import sys
import time

sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes  # noqa: E402

W, H, B = 1920, 1080, 8
for name, spec, b in (("cornell", scenes.cornell_box(), 8), ("mesh100k", scenes.displaced_grid_mesh(100000), 8)):
    for n in (1, 2, 4, 8, 16, 32):
        row = []
        for schedule in (0, 1, -1):
            sc = api.Scene(W, H).load(spec)
            sc.set_option("schedule", schedule)
            sc.iterations = b
            for _ in range(2):
                sc.render(n)
            sc.sync()
            reps = max(2, 32 // n)
            t = time.time()
            for _ in range(reps):
                sc.render(n)
            sc.sync()
            dt = time.time() - t
            row.append(W * H * n * reps / dt / 1e6)
        print("%s render(%2d): lockstep %7.1f  suspend %7.1f  auto %7.1f Msamples/s" % (name, n, row[0], row[1], row[2]), flush=True)
