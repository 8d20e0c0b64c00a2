#!/bin/bash
# round 3, job h: wavefront variant after the rework (lazy per-material path state, dynamic ray ranges): tests, rate, per-kernel trace + counters
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "wavefront or closed_form or mesh_configs or schedules or variant" 2>&1 | tail -3
python3 - <<'PY'
import sys, time
sys.path.insert(0, ".")
from opencl_path_tracer_amd import api, scenes
for name, spec, b, spp in (("cornell", scenes.cornell_box(), 8, 16), ("mesh100k", scenes.displaced_grid_mesh(100000), 8, 8)):
    for lib_opts in ({}, {"cost_binning": 0}):
        sc = api.Scene(1920, 1080).load(spec)
        sc.set_option("variant", 1)
        for k, v in lib_opts.items():
            sc.set_option(k, v)
        sc.iterations = b
        sc.render(2); sc.sync()
        t = time.time(); sc.render(spp); sc.sync(); dt = time.time() - t
        print("%s wavefront %s: %.1f Msamples/s (%.3f ms per sample pass)" % (name, lib_opts, 1920 * 1080 * spp / dt / 1e6, dt / spp * 1e3), flush=True)
PY
for sc in cornell mesh100k; do
  rocprofv3 --kernel-trace --stats -d gpurun_out/r3h_wf_${sc}_trace -o t --output-format csv -- python3 tools/prof_render.py scene=$sc spp=8 reps=2 variant=1 > gpurun_out/r3h_wf_${sc}_trace.log 2>&1 || { echo "wf trace $sc failed"; exit 1; }
  python3 tools/trace_summary.py gpurun_out/r3h_wf_${sc}_trace > gpurun_out/r3h_wf_${sc}_trace_summary.txt; cat gpurun_out/r3h_wf_${sc}_trace_summary.txt
  tools/pmc_sets.sh r3h_wf_$sc "FETCH_SIZE;WRITE_SIZE;SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" scene=$sc spp=8 reps=2 variant=1 > gpurun_out/r3h_wf_$sc.log 2>&1 || { echo "wf pmc $sc failed"; exit 1; }
  grep -E "FETCH_SIZE|WRITE_SIZE" gpurun_out/pmc_r3h_wf_${sc}_summary.txt
done
echo done
