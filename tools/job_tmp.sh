cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; out=gpurun_out/r4q; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?; tail -5 $out/tests.log; [ $rc -ne 0 ] && exit 1
python - > $out/wf_streams.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
for name, make, b, spp in (("cornell", scenes.cornell_box, 8, 8), ("mesh100k", lambda: scenes.displaced_grid_mesh(100000), 8, 4), ("mesh1m", lambda: scenes.displaced_grid_mesh(1000000), 16, 2)):
    sc = api.Scene(1920, 1080).load(make())
    sc.iterations = b
    sc.set_option("variant", 1)
    for ns in (2, 4, 6, 8):
        sc.set_option("wf_streams", ns)
        sc.render(spp); sc.sync()
        t = time.time()
        for _ in range(3): sc.render(spp)
        sc.sync()
        print("%-9s wf_streams %d: %8.1f Msamples/s" % (name, ns, 1920 * 1080 * spp * 3 / (time.time() - t) / 1e6), flush=True)
    sc.close()
PY
cat $out/wf_streams.txt
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -3 $out/bench.err; cut -c1-600 $out/bench.json
