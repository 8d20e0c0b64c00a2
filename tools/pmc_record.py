"""Turn the rocprofv3 --pmc passes of tools/pmc_passes.sh into the tracked summary bench.py reads.

  python tools/pmc_record.py <key> <out.json> <pmc dirs...>        e.g.
  python tools/pmc_record.py cornell_1920x1080_b8_spp64 profiles/counters.json gpurun_out/pmc_cb_*

Per key: VALU instructions per launch, active-lane fraction (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)),
wave-cycle split, L1 / L2 hit rates, fabric-side bytes per launch (FETCH_SIZE x factor + WRITE_SIZE, both in KiB; the factor is
2 for coalesced streams -- the counter tallies 128-B requests as 64 B, MI355X_MICROARCH.md -- and 1 for the traversal's 64-byte
gathers, calibrated in profiles/r04/g_*), the shader clock during the profiled
launches (GRBM_GUI_ACTIVE / 8 XCDs / kernel time) and a hash of the kernel sources they were measured with."""
import csv
import glob
import json
import os
import subprocess
import sys
import time
from collections import defaultdict

sys.path.insert(0, ".")
from bench import kernel_source_sha  # noqa: E402

args = sys.argv[1:]
fetch_factor = None
if "--fetch-factor" in args:
    i = args.index("--fetch-factor")
    fetch_factor = float(args[i + 1])
    del args[i:i + 2]
sys.argv = [sys.argv[0]] + args
key, out = sys.argv[1], sys.argv[2]
# FETCH_SIZE tallies 64 B per request of the L2 towards the fabric (MI355X_MICROARCH.md): a wide coalesced read asks for 128-B
# lines (x2), the traversal's 64-byte per-lane gathers for 64-B sectors (x1: profiles/r04/g_fetch_size_calibration_64B_gathers.txt).
# The megakernel on a scene read from global memory is all gathers, on the Cornell box (tree in LDS) all streaming; the wavefront
# variant streams.  --fetch-factor overrides.
if fetch_factor is None:
    fetch_factor = 1.0 if key.startswith("mesh") else 2.0
wavefront = key.startswith("wavefront")
vals, durs = defaultdict(list), []
if wavefront:
    # (profile it with wf_streams=1: one chain, so that a wf_generate launch IS a sample pass of the whole frame; under the
    # profiler the kernels of several chains are serialised anyway)
    # the wavefront variant: a "launch" is one sample pass = wf_generate + iterations x (wf_intersect + wf_shade); counters
    # and kernel time are summed over all of them and divided by the number of passes (= wf_generate launches) of the run
    tot, npass, tdur = defaultdict(float), defaultdict(int), defaultdict(float)
    for d in sys.argv[3:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "wf_" in r["Kernel_Name"]:
                    tot[(d, r["Counter_Name"])] += float(r["Counter_Value"])
                    if "wf_generate" in r["Kernel_Name"]:
                        npass[(d, r["Counter_Name"])] += 1
        for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "wf_" in r["Kernel_Name"]:
                    tdur[d] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                    if "wf_generate" in r["Kernel_Name"]:
                        npass[(d, "_trace")] += 1
    for (d, cn), v in tot.items():
        vals[cn].append(v / max(npass[(d, cn)], 1))
    durs = [tdur[d] / max(npass[(d, "_trace")], 1) for d in tdur]
else:
    for d in sys.argv[3:]:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_render" in r["Kernel_Name"]:
                    vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_render" in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)


def steady(v):          # drop the first (warm-up) launch of every pass when there are several
    if wavefront:
        return sum(v) / len(v)
    return sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]


c = {k: steady(v) for k, v in vals.items()}
dur = sorted(durs)[len(durs) // 2]
entry = {
    "kernel": "wf_generate + wf_intersect + wf_shade, one sample pass" if wavefront else "k_render", "launch_seconds_under_profiler": dur,
    "valu_insts_per_launch": c.get("SQ_INSTS_VALU"),
    "active_lane_fraction": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if "SQ_THREAD_CYCLES_VALU" in c else None,
    "shader_clock_hz": c["GRBM_GUI_ACTIVE"] / 8.0 / dur if "GRBM_GUI_ACTIVE" in c else 2.4e9,
    "wave_cycles": {k: c.get(k) for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if "TCC_HIT_sum" in c else None,
    "l1_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"] if "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None,
    "hbm_bytes_per_launch": (fetch_factor * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
    "fetch_size_factor": fetch_factor,
    "hbm_note": ("L2-miss traffic towards the fabric: FETCH_SIZE x %.0f (calibrated for this access pattern, profiles/r04/g_*) + WRITE_SIZE. "
                 "Infinity-Cache hits are counted in it: the scene (nodes + packets + meta) is %s" % (
                     fetch_factor, "72 MB (MESH-1M) / 7 MB (MESH-100k), resident in the 256-MiB Infinity Cache after first touch, so the HBM share is "
                     "the compulsory part only (scene once + the frame buffers per launch)" if key.startswith("mesh") else "LDS / L2 resident")),
    "fetch_kib_reported": c.get("FETCH_SIZE"), "write_kib": c.get("WRITE_SIZE"),
    "counters": c,
    "kernel_source_sha": kernel_source_sha(),
    "measured_at": time.strftime("%Y-%m-%d %H:%M:%S"),
    "commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None,
    "source": "rocprofv3 --kernel-trace --pmc <set> -- python3 tools/prof_render.py (tools/pmc_passes.sh), steady-state launches",
}
allj = json.load(open(out)) if os.path.exists(out) else {}
allj[key] = entry
json.dump(allj, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in entry.items() if k != "counters"}, indent=1))
