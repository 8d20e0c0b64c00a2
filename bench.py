#!/usr/bin/env python3
"""bench.py -- headline benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s, whole job, at 1920x1080, 8 bounces, Cornell box
(config 2: 1,024 spp at the default --steps 16 x 64 spp).  A sample is one camera path of up
to 8 segments.  One "step" = one pass of the hot path (gen_ray + trace_ray, fused) taking
--spp-per-step samples of every pixel of the frame.

N > 1 (launched by torch.distributed.run, one process per GPU): the frame is tiled in
interleaved 8-row blocks over the ranks (strong scaling: the frame is fixed), every rank
renders its own pixels with no communication, and the timed region ends with ONE RCCL
all-gather of the radiance slabs + de-interleave on every rank -- issued through the
library's own C ABI (pt_comm_init / pt_gather_frame: ncclAllGather on the render stream + a
de-interleave kernel); torch.distributed only launches the processes, carries the 128-byte
communicator id and reduces the timing scalars.  --exchange torch uses round 1's
all_gather_into_tensor + index_copy_ instead.

The timed region starts with scene, seeds and framebuffer resident in HBM and is bracketed by
barrier + torch.cuda.synchronize(); the time is the max over ranks.  Rank 0 prints ONE JSON line.

Extra objects:
  "roofline"       SURVEY 8(d)'s algorithmic HBM bytes of the kernel that runs / HIP-event kernel time vs
                   the 8 TB/s peak.  Megakernel (default): 40 B/sample (colors 16 B R + 16 B W, rnds 4 B R +
                   4 B W) -- the path state never leaves the registers, so this fraction is tiny and says
                   only that the kernel is NOT HBM bound; wavefront: 32 B/sample + 200 B/segment.
  "roofline_valu"  what bounds the megakernel: VALU issue.  VALU instructions per SIMD-cycle (peak 0.5: one
                   wave64 instruction per 2 cycles) and the active-lane fraction, from the tracked rocprofv3
                   SQ-counter summary (profiles/counters.json, written by tools/pmc_record.py); "stale" tells
                   whether the kernel sources changed since it was measured.  instructions-per-launch x live
                   launch count / live kernel time is re-derived with THIS run's HIP-event time.
  "cpu_baseline"   the CPU oracle (port of the reference path) timed on this host's cores on a bounded
                   sample of the same workload; rank 0, N = 1 only.
  "configs"        the other BASELINE.json configs on this GPU (N = 1 only, untimed side runs after the headline's timed
                   region, which they do not touch): config 1 (Cornell box 256x256, 4 bounces, 16 spp), config 3 (MESH-100k
                   through pt_add_obj, 1080p, 8 bounces), config 4 at N = 1 (Cornell box 3840x2160) and config 5 (MESH-1M
                   through pt_add_obj, 1080p, 16 bounces); each with ms_per_step, msamples_per_s, mean_path_segments, its
                   40-B/sample HBM roofline, and roofline_valu / traffic from its profiles/counters.json entry.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (first: libptamd.so must bind to the HIP runtime torch loaded)
import torch.distributed as dist  # noqa: E402

from opencl_path_tracer_amd import api, scenes  # noqa: E402

WIDTH, HEIGHT, BOUNCES = 1920, 1080, 8
ROWS_PER_BLOCK = 8
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E (spec)
BYTES_PER_SEGMENT = 200.0        # SURVEY 8(d), wavefront: 92 B state R + 92 B W + 8 B hit W + 8 B R
BYTES_PER_SAMPLE = 32.0          # SURVEY 8(d), wavefront: colors 16 B R + 16 B W
MEGA_BYTES_PER_SAMPLE = 40.0     # SURVEY 8(d), megakernel: colors RMW + seed RMW
VALU_PEAK_PER_SIMD_CYCLE = 0.5   # MI355X_MICROARCH.md: v_fma_f32 (wave64) 2 cycles on a SIMD
N_SIMD = 1024                    # 256 CUs x 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--bounces", type=int, default=BOUNCES)
    ap.add_argument("--lds-scene", type=int, default=-1, help="-1 library default, 0 nodes through L1/L2, 2 staged in LDS")
    ap.add_argument("--exchange", choices=["cabi", "torch"], default="cabi", help="N > 1: frame assembly through pt_gather_frame (default) or torch.distributed")
    ap.add_argument("--variant", type=int, default=0, help="0 megakernel (default, fastest), 1 wavefront (stream-compacted)")
    ap.add_argument("--no-variants", action="store_true", help="skip the untimed side measurement of the other variant")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all ranks share cuda:0 and the exchange runs over gloo on host copies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--collective-timeout", type=float, default=180.0, help="N > 1: seconds any rendezvous / collective phase may take before the rank exits non-zero")
    ap.add_argument("--no-band-check", action="store_true", help="skip the one-rank band render that proves the frame equals an N = 1 render")
    ap.add_argument("--no-configs", action="store_true", help="skip the side runs of BASELINE configs 1, 3, 4 (N = 1) and 5")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(width, height, bounces, target_s):
    """The oracle (C port of prog.cl's gen_ray/trace_ray incl. the reference's own kd-tree
    traversal) on all host cores, bounded sample of the same workload.  Built here, for THIS host
    (oracle/Makefile target `native`: -O3 -march=native); if that build fails the -O2 library
    the tests use is timed instead, and `sample` says which."""
    import subprocess
    import tempfile
    flags = "-O2 -mfma -ffp-contract=off (oracle/libpt_oracle.so: the native build failed)"
    tmp = tempfile.mkdtemp(prefix="ptamd_oracle_")
    so = os.path.join(tmp, "libpt_oracle_native.so")
    try:
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", "NATIVE_OUT=" + so], check=True, capture_output=True, timeout=120)
        os.environ["PT_ORACLE_LIB"] = so
        flags = "gcc -O3 -march=native -ffp-contract=off, built on this host"
    except (OSError, subprocess.SubprocessError):
        pass
    from oracle import oracle_py as O
    spec = scenes.cornell_box()
    osc = O.load_scene(spec)
    cam = O.make_camera(spec.fov, spec.yaw, spec.pitch, spec.shift, width, height)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # calibrate on a 1/16-area strip: rows [0, height/16) via a short frame of the same width
    fr = O.OracleFrame(width, height)
    t0 = time.time()
    # one sample of the whole frame is the smallest unit that keeps the workload's pixel mix
    fr.render(osc, cam, bounces, 0, 1, mode=0, nthreads=cores)
    t1 = time.time() - t0
    spp = 1
    extra = int(max(0, min(15, (target_s - t1) // max(t1, 1e-3))))
    if extra > 0:
        t0 = time.time()
        fr.render(osc, cam, bounces, 1, extra, mode=0, nthreads=cores)
        t1 += time.time() - t0
        spp += extra
    samples = width * height * spp
    return {"value": samples / t1 / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "flags": flags,
            "sample": "Cornell box %dx%d, %d bounces, %d spp (full frame), oracle/pt_oracle.c mode 0 (%s), %.1f s" % (width, height, bounces, spp, flags, t1)}


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    # device code AND the host code that picks the kernel instance and its launch shape (schedule, pass length, waves per
    # SIMD, node layout): a change in either invalidates counters measured under the old one
    # ... and the builders of the tree the mesh counters are measured on, and the compiler flags
    for f in ("pt_device.hpp", "pt_kernels.hip", "pt_wavefront.hip", "pt_internal.hpp", "pt_context.hpp", "pt_host.cpp", "pt_builder.cpp", "pt_launch.cpp", "pt_wide.cpp", "pt_sahdev.hip", "pt_widedev.hip"):
        h.update(open(os.path.join(ROOT, "opencl_path_tracer_amd", "csrc", f), "rb").read())
    h.update(open(os.path.join(ROOT, "Makefile"), "rb").read())
    return h.hexdigest()[:16]


def tracked_counters(W, H, B, spp, prefix="cornell"):
    """rocprofv3 PMC summary of the same launch shape (tools/pmc_record.py), or None.  `stale` = the kernel
    sources are not the ones it was measured with."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        tj = json.load(open(path))
    except (OSError, ValueError):
        return None
    c = tj.get("%s_%dx%d_b%d_spp%d" % (prefix, W, H, B, spp))
    if not c:
        return None
    c = dict(c)
    c["stale"] = c.get("kernel_source_sha") != kernel_source_sha()
    c["source"] = "profiles/counters.json <- " + c.get("source", "?")
    return c


def valu_roofline(counters, mean_launch_ms):
    """VALU-issue roofline from a tracked counter entry and a live kernel time."""
    ipc = counters["valu_insts_per_launch"] / N_SIMD / (mean_launch_ms * 1e-3 * counters["shader_clock_hz"])
    return {"bound": "valu_issue", "achieved": ipc, "peak": VALU_PEAK_PER_SIMD_CYCLE,
            "unit": "VALU instructions per SIMD-cycle", "frac": ipc / VALU_PEAK_PER_SIMD_CYCLE,
            "active_lane_fraction": counters["active_lane_fraction"],
            "valu_insts_per_launch": counters["valu_insts_per_launch"],
            "l2_hit_rate": counters.get("l2_hit_rate"),
            "source": counters["source"], "measured_at": counters["measured_at"], "stale": counters["stale"]}


def mesh_scene_through_add_obj(ntris, W, H, device, workdir):
    """BASELINE configs 3 / 5 as they are worded: the Cornell walls authored with add_Triangle (main.cpp:793-815) and the
    mesh written as OBJ+MTL (Kd/Ks/Ke/Ns/Kn/Kk/Tp) and loaded with pt_add_obj (main.cpp:552-617).  Identity transform:
    the loader then authors exactly the triangles of scenes.displaced_grid_mesh(), the scene profiles/counters.json was
    measured on.  Returns (scene, seconds spent in pt_add_obj, seconds in upload_Triangles)."""
    path, _, _, _ = scenes.write_grid_mesh_obj(ntris, workdir)
    sc = api.Scene(W, H, device=device)
    for m in scenes.BUILTIN_MATERIALS:
        sc.add_Material(*m)
    wv, wm = scenes.cornell_walls()
    sc.add_Triangles(api.triangles_from_vertices(wv, wm))
    sc.end_Obj()
    t0 = time.perf_counter()
    sc.add_Obj(path, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0.0, 0.0)
    t1 = time.perf_counter()
    sc.upload_Triangles()
    sc.sync()
    t2 = time.perf_counter()
    sc.upload_Materials()
    sc.set_view(60.0, 0.0, 0.0, (0.0, 0.0, 0.0))
    return sc, t1 - t0, t2 - t1


def run_config(name, workload, make_scene, W, H, B, spp, steps, counters_key):
    """One BASELINE config on this GPU: `steps` timed launches of `spp` samples after one warm-up launch."""
    made = make_scene()
    sc, extra = (made[0], {"add_obj_ms": made[1] * 1e3, "upload_triangles_ms": made[2] * 1e3}) if isinstance(made, tuple) else (made, {})
    sc.iterations = B
    sc.set_option("timing", 1)
    sc.render(spp)
    sc.sync()
    sc.current_sample = 0
    sc.seed_default()
    sc.set_option("reset_stats", 1)
    sc.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.render(spp)
    sc.sync()
    dt = time.perf_counter() - t0
    segs, samples, kms, launches = sc.stat("segments"), sc.stat("samples"), sc.stat("kernel_ms"), sc.stat("kernel_launches")
    assert samples == float(W) * H * spp * steps, (name, samples)
    mean_launch_ms = kms / launches
    bytes_per_launch = MEGA_BYTES_PER_SAMPLE * (samples / launches)
    achieved = bytes_per_launch / (mean_launch_ms * 1e-3) / 1e9
    c = tracked_counters(W, H, B, spp, prefix=counters_key) if counters_key else None
    out = {"config": name, "workload": workload, "steps": steps, "spp_per_step": spp, "ms_per_step": dt / steps * 1e3,
           "msamples_per_s": samples / dt / 1e6, "mean_path_segments": segs / samples, "msegments_per_s": segs / dt / 1e6,
           "kernel": {"name": "k_render", "node_mode": int(sc.stat("node_mode")), "waves_per_simd": int(sc.stat("waves_per_simd")),
                      "bvh_nodes": int(sc.stat("bvh_nodes")), "lds_bytes": int(sc.stat("lds_bytes")), "mean_launch_ms": mean_launch_ms},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": (c["hbm_bytes_per_launch"] if c and not c["stale"] else None),
                        "traffic_note": (c.get("hbm_note") if c else None),
                        "algorithmic_bytes_per_launch": bytes_per_launch, "model": "SURVEY 8(d): megakernel 40 B/sample"},
           "roofline_valu": valu_roofline(c, mean_launch_ms) if c else None}
    out.update(extra)
    sc.close()
    return out


def all_configs(device):
    """BASELINE.json configs[0], [2], [3] (at N = 1) and [4]; configs[1] is the headline itself.  The mesh configs take 64 samples
    per step like the headline (samples per launch are the caller's choice -- render(n) -- and a launch's ragged end weighs less the
    longer it is: MESH-1M 374 / 409 / 420 Msamples/s at 8 / 32 / 64 spp per launch, profiles/r04/j_*)."""
    import shutil
    import tempfile
    cb = scenes.cornell_box
    work = tempfile.mkdtemp(prefix="ptamd_bench_")
    try:
        return [
            run_config("config 1", "Cornell box, 256x256, 4 bounces, 16 spp per step (the reference CPU path's config: 16 spp in all)", lambda: api.Scene(256, 256, device=device).load(cb()), 256, 256, 4, 16, 16, "cornell"),
            run_config("config 3", "MESH-100k (OBJ+MTL through pt_add_obj, 100,352 + 12 triangles), 1920x1080, 8 bounces, 64 spp per step",
                       lambda: mesh_scene_through_add_obj(100000, 1920, 1080, device, work), 1920, 1080, 8, 64, 2, "mesh100k"),
            run_config("config 4 at N=1", "Cornell box, 3840x2160, 8 bounces, 16 spp per step", lambda: api.Scene(3840, 2160, device=device).load(cb()), 3840, 2160, 8, 16, 3, "cornell"),
            run_config("config 5", "MESH-1M (OBJ+MTL through pt_add_obj, 1,002,528 + 12 triangles), 1920x1080, 16 bounces, 64 spp per step",
                       lambda: mesh_scene_through_add_obj(1000000, 1920, 1080, device, work), 1920, 1080, 16, 64, 2, "mesh1m"),
        ]
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    gpu_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    comm_dev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
    from opencl_path_tracer_amd.distributed import TileMap, Watchdog, band_context, band_pixel_ids, exchange_frame, frame_matches_band

    def guarded(what):          # every phase that can block on another rank is bounded: a missing rank fails the job
        return Watchdog(args.collective_timeout if world > 1 else 0.0, what, rank)

    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with guarded("torch.distributed rendezvous"):
            if args.rehearse_on_one_gpu:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.collective_timeout))
            else:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=args.collective_timeout))

    W, H, B = args.width, args.height, args.bounces
    spec = scenes.cornell_box()
    sc = api.Scene(W, H, device=gpu_index, rank=rank, world=world, rows_per_block=ROWS_PER_BLOCK).load(spec)
    sc.iterations = B
    if args.lds_scene >= 0:
        sc.set_option("lds_scene", args.lds_scene)
    sc.set_option("variant", args.variant)
    sc.set_option("timing", 1)

    # device memory and stream are torch's: the radiance slab is a torch tensor so that RCCL
    # (torch.distributed) can gather it; padded to the largest rank's pixel count.
    tmap = TileMap(W, H, world, ROWS_PER_BLOCK)
    npix = sc.local_pixels
    assert npix == tmap.count(rank)
    slab = torch.zeros((tmap.max_count, 4), dtype=torch.float32, device=dev)
    rnds = torch.zeros((tmap.max_count,), dtype=torch.int32, device=dev)
    sc.bind_framebuffer(slab.data_ptr(), rnds.data_ptr())
    stream = torch.cuda.current_stream(dev)
    sc.set_stream(stream.cuda_stream)
    gathered = torch.empty((world * tmap.max_count, 4), dtype=torch.float32, device=comm_dev) if world > 1 else None
    frame = torch.empty((W * H + 1, 4), dtype=torch.float32, device=comm_dev) if world > 1 else None
    scatter_index = tmap.gather_index(comm_dev) if world > 1 else None

    def sync_all(what="barrier + synchronize"):
        with guarded(what):
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

    use_cabi = world > 1 and args.exchange == "cabi" and not args.rehearse_on_one_gpu
    exchange_note = None
    if use_cabi:          # the library's own RCCL communicator; torch only carries the 128-byte id
        # Every rank must take the same path, and a rank that cannot take part must be found BEFORE anyone enters
        # ncclCommInitRank (the others would block in it): (1) every rank probes the RCCL binding (pt_comm_available: no
        # collective), (2) the flags are MIN-reduced, (3) only if all can, rank 0 creates the id and all ranks call
        # pt_comm_init -- under the watchdog, like every phase that waits for another rank.  Any failure switches ALL ranks
        # to torch's all-gather, loudly: stderr, and "exchange" / "exchange_note" in the JSON line say which path ran.
        ok, why = api.comm_available()
        ok = 1 if ok else 0
        flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
        with guarded("capability all_reduce"):
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            box = [None]
            if rank == 0:
                try:
                    box = [api.comm_unique_id()]
                except api.PtError as e:
                    why = "pt_comm_unique_id: %s" % e
            with guarded("broadcast of the communicator id"):
                dist.broadcast_object_list(box, src=0)
            if box[0] is None:
                ok, why = 0, why or "rank 0 could not create a communicator id"
            else:
                try:
                    with guarded("pt_comm_init (ncclCommInitRank)"):
                        sc.comm_init(box[0])
                except api.PtError as e:
                    ok, why = 0, "pt_comm_init: %s" % e
            flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
            with guarded("communicator all_reduce"):
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            use_cabi = False
            exchange_note = "C-ABI exchange unavailable (%s): torch.distributed all_gather_into_tensor used instead" % (why or "another rank failed")
            print("[bench] rank %d: %s" % (rank, exchange_note), file=sys.stderr, flush=True)

    def exchange():
        if use_cabi:
            sc.gather_frame()           # ncclAllGather + de-interleave kernel on the render stream
            return None
        src = slab.to(comm_dev) if (world > 1 and comm_dev != dev) else slab
        return exchange_frame(src, tmap, scatter_index, gathered, frame)

    # ---- warmup (untimed)
    for _ in range(args.warmup):
        sc.render(args.spp_per_step)
    exchange()
    sync_all()
    sc.current_sample = 0
    sc.seed_default()
    sc.set_option("reset_stats", 1)

    # ---- timed region: exactly K steps + the final exchange
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.render(args.spp_per_step)
    out = exchange()
    sync_all("timed region: render + exchange")
    dt = time.perf_counter() - t0
    timed_stats = (sc.stat("segments"), sc.stat("samples"), sc.stat("kernel_ms"), sc.stat("kernel_launches"))

    # ---- the exchange alone (untimed repeat of the final step of the timed region: same buffers, same result)
    exchange_ms = None
    if world > 1:
        sync_all()
        t1 = time.perf_counter()
        out = exchange()
        sync_all("exchange alone")
        exchange_ms = (time.perf_counter() - t1) * 1e3

    # ---- self-validation: the frame of an N-rank run is bit-identical to a one-rank render by construction (seeds and
    # pixel ids are those of the global frame) -- so prove it: rank 0 renders a 24-row band as a ONE-rank context, same
    # seeds, same spp, and compares it bit for bit with those rows of the assembled frame
    frame_matches_n1, band = None, None
    if rank == 0 and not args.no_band_check:
        band = band_context(H, 24)
        frame_np = sc.read_frame() if (use_cabi or world == 1) else out.detach().cpu().numpy()
        bsc = api.Scene(W, H, device=gpu_index, **band).load(spec)
        bsc.iterations = B
        if args.lds_scene >= 0:
            bsc.set_option("lds_scene", args.lds_scene)
        bsc.set_option("variant", args.variant)
        for _ in range(args.steps):
            bsc.render(args.spp_per_step)
        frame_matches_n1 = frame_matches_band(frame_np, bsc.read_colors(), band_pixel_ids(W, H, band))
        bsc.close()
        if not frame_matches_n1:
            print("[bench] the assembled frame DIFFERS from a one-rank render of rows %d..%d" % (band["rank"] * 24, band["rank"] * 24 + 23), file=sys.stderr, flush=True)

    # ---- side measurement (untimed, not part of `value`): the other formulation, same workload
    other = None
    if world == 1 and not args.no_variants:
        ov = 1 - args.variant
        sc.set_option("variant", ov)
        sc.render(args.spp_per_step)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        nside = max(1, min(8, args.steps // 4))
        for _ in range(nside):
            sc.render(args.spp_per_step)
        torch.cuda.synchronize(dev)
        side_dt = time.perf_counter() - t1
        other = {"variant": "wavefront" if ov == 1 else "megakernel",
                 "msamples_per_s": W * H * args.spp_per_step * nside / side_dt / 1e6, "ms_per_step": side_dt / nside * 1e3}
        sc.set_option("variant", args.variant)

    tmax = torch.tensor([dt], dtype=torch.float64, device=comm_dev)
    stats = torch.tensor(list(timed_stats), dtype=torch.float64, device=comm_dev)
    kmax, kmin = stats[2:3].clone(), stats[2:3].clone()
    xmax = torch.tensor([exchange_ms or 0.0], dtype=torch.float64, device=comm_dev)
    if world > 1:
        with guarded("reduction of the timing scalars"):
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(kmin, op=dist.ReduceOp.MIN)
            dist.all_reduce(xmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    segs, samples, kms_sum, launches = [float(x) for x in stats.tolist()]
    total_spp = args.steps * args.spp_per_step
    assert samples == float(W) * H * total_spp, (samples, W * H * total_spp)
    if use_cabi or world == 1:
        checksum = float(sc.read_frame()[:, :3].astype(np.float64).sum())
    else:
        checksum = float(out[:, :3].double().sum().item())

    if rank == 0:
        dbar = segs / samples
        value = samples / dt / 1e6
        # dominant kernel: k_render (or wf_intersect).  Algorithmic bytes per launch / mean launch time.
        launches_per_rank = launches / world
        wavefront_model_bytes = (BYTES_PER_SAMPLE + BYTES_PER_SEGMENT * dbar) * (samples / launches)
        if args.variant == 0:
            bytes_per_launch = MEGA_BYTES_PER_SAMPLE * (samples / launches)
        else:   # wf_intersect alone: 32 B ray read + 8 B hit record written per ray of the launch
            bytes_per_launch = 40.0 * (segs / launches)
        mean_launch_ms = (kms_sum / world) / launches_per_rank
        achieved = bytes_per_launch / (mean_launch_ms * 1e-3) / 1e9
        counters = tracked_counters(W, H, B, args.spp_per_step) if world == 1 and args.variant == 0 else None
        traffic = counters["hbm_bytes_per_launch"] if counters and not counters["stale"] else None
        line = {
            "metric": "Msamples/s (whole node) at 1920x1080, 8-bounce Cornell box",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Cornell box (12 wall/lamp triangles + 2 tessellated spheres = 1,932 triangles), "
                                   "%dx%d, %d bounces, %d spp (%d steps x %d spp)" % (W, H, B, total_spp, args.steps, args.spp_per_step),
                       "parallelism": "tiles%d" % world, "rows_per_block": ROWS_PER_BLOCK,
                       "exchange": None if world == 1 else ("pt_gather_frame (ncclAllGather through the C ABI)" if use_cabi else "torch.distributed all_gather_into_tensor"),
                       "exchange_note": exchange_note,
                       "variant": "megakernel" if args.variant == 0 else "wavefront",
                       "kernel": "k_render (fused gen_ray+trace_ray, persistent per pixel)" if args.variant == 0 else "wf_intersect (+wf_generate, wf_shade)"},
            "mean_path_segments": dbar, "msegments_per_s": segs / dt / 1e6, "radiance_checksum": checksum,
            # N-rank self-validation (also printed, trivially, at N = 1): a 24-row band rendered by a one-rank context on rank 0
            # == the same rows of the assembled frame, bit for bit; which exchange ran; the slowest and the fastest rank's kernel
            # time over the timed steps; the exchange alone (max over ranks, untimed repeat)
            "frame_matches_n1": frame_matches_n1,
            "band_check": None if band is None else {"rows": [band["rank"] * band["rows_per_block"], min(H, (band["rank"] + 1) * band["rows_per_block"]) - 1],
                                                      "context": "pt_create_tiled(rank=%d, world=%d, rows_per_block=%d)" % (band["rank"], band["world"], band["rows_per_block"])},
            "exchange_path": "none (one rank: the colors buffer is the frame)" if world == 1 else ("pt_gather_frame" if use_cabi else ("gloo all_gather (rehearsal)" if args.rehearse_on_one_gpu else "torch all_gather_into_tensor")),
            "kernel_ms_per_rank": {"min": float(kmin.item()), "max": float(kmax.item())},
            "exchange_ms": (float(xmax.item()) if world > 1 else 0.0),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_render" if args.variant == 0 else "wf_intersect", "mean_launch_ms": mean_launch_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "model": "SURVEY 8(d): megakernel 40 B/sample (path state stays in registers)" if args.variant == 0 else "wf_intersect: 40 B/ray",
                         "wavefront_model_frac": wavefront_model_bytes / (mean_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if counters is not None:
            # the bound the megakernel actually has: VALU issue, with this run's kernel time
            line["roofline_valu"] = valu_roofline(counters, mean_launch_ms)
        if other is not None:
            # the other formulation's own HBM roofline.  Wavefront: SURVEY 8(d)'s stream model, 32 B/sample + 200 B/segment
            # over ALL its kernels (generate, intersect, shade) per step; traffic = the tracked PMC bytes of one sample pass
            # x spp (profiles/counters.json key wavefront_*, tools/pmc_record.py), null when stale or absent.
            if other["variant"] == "wavefront":
                obytes = (BYTES_PER_SAMPLE + BYTES_PER_SEGMENT * dbar) * W * H * args.spp_per_step
                omodel = "SURVEY 8(d): 32 B/sample + 200 B/segment, all wavefront kernels of a step"
                oc = tracked_counters(W, H, B, 1, prefix="wavefront_cornell")
                otraffic = oc["hbm_bytes_per_launch"] * args.spp_per_step if oc and not oc["stale"] else None
            else:
                obytes = MEGA_BYTES_PER_SAMPLE * W * H * args.spp_per_step
                omodel = "SURVEY 8(d): megakernel 40 B/sample"
                otraffic = None
            oach = obytes / (other["ms_per_step"] * 1e-3) / 1e9
            other["roofline"] = {"bound": "hbm", "achieved": oach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": oach / HBM_PEAK_GBS,
                                 "traffic": otraffic, "algorithmic_bytes_per_step": obytes, "model": omodel,
                                 "time": "wall clock of the untimed side run (all launches of a step)"}
            line["other_variant"] = other
        if world == 1 and not args.no_configs:
            sc.close()
            line["configs"] = all_configs(gpu_index)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(W, H, B, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        with guarded("final barrier"):
            dist.barrier()
        dist.destroy_process_group()
    if frame_matches_n1 is False:
        raise SystemExit(5)


if __name__ == "__main__":
    main()
