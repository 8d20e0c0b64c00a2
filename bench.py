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
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def cpu_baseline(width, height, bounces, target_s):
    """The oracle (C port of prog.cl's gen_ray/trace_ray incl. the reference's own kd-tree
    traversal) on all host cores, bounded sample of the same workload."""
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
    return {"value": samples / t1 / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "Cornell box %dx%d, %d bounces, %d spp (full frame), oracle/pt_oracle.c mode 0, %.1f s" % (width, height, bounces, spp, t1)}


def kernel_source_sha():
    import hashlib
    h = hashlib.sha256()
    # device code AND the host code that picks the kernel instance and its launch shape (schedule, pass length, waves per
    # SIMD, node layout): a change in either invalidates counters measured under the old one
    # ... and the builders of the tree the mesh counters are measured on, and the compiler flags
    for f in ("pt_device.hpp", "pt_kernels.hip", "pt_wavefront.hip", "pt_internal.hpp", "pt_host.cpp", "pt_wide.cpp", "pt_sahdev.hip", "pt_widedev.hip"):
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

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
    from opencl_path_tracer_amd.distributed import TileMap, exchange_frame
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

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    use_cabi = world > 1 and args.exchange == "cabi" and not args.rehearse_on_one_gpu
    exchange_note = None
    if use_cabi:          # the library's own RCCL communicator; torch only carries the 128-byte id
        # Every rank must take the same path: a rank that cannot bind RCCL / create the communicator says so, the flags are
        # reduced, and if ANY rank failed ALL ranks use torch's all-gather instead -- loudly: stderr, and "exchange" /
        # "exchange_note" in the JSON line say which path produced the number.
        ok, why = 1, ""
        try:
            box = [api.comm_unique_id() if rank == 0 else None]
        except api.PtError as e:
            box, ok, why = [None], 0, "pt_comm_unique_id: %s" % e
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            ok, why = 0, why or "rank 0 could not create a communicator id"
        if ok:
            try:
                sc.comm_init(box[0])
            except api.PtError as e:
                ok, why = 0, "pt_comm_init: %s" % e
        flag = torch.tensor([ok], dtype=torch.int32, device=comm_dev)
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
    sync_all()
    dt = time.perf_counter() - t0
    timed_stats = (sc.stat("segments"), sc.stat("samples"), sc.stat("kernel_ms"), sc.stat("kernel_launches"))

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
    kmax = stats[2:3].clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    segs, samples, kms_sum, launches = [float(x) for x in stats.tolist()]
    total_spp = args.steps * args.spp_per_step
    assert samples == float(W) * H * total_spp, (samples, W * H * total_spp)
    if use_cabi:
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
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_render" if args.variant == 0 else "wf_intersect", "mean_launch_ms": mean_launch_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "model": "SURVEY 8(d): megakernel 40 B/sample (path state stays in registers)" if args.variant == 0 else "wf_intersect: 40 B/ray",
                         "wavefront_model_frac": wavefront_model_bytes / (mean_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        if counters is not None:
            # the bound the megakernel actually has: VALU issue, with this run's kernel time
            ipc = counters["valu_insts_per_launch"] / N_SIMD / (mean_launch_ms * 1e-3 * counters["shader_clock_hz"])
            line["roofline_valu"] = {"bound": "valu_issue", "achieved": ipc, "peak": VALU_PEAK_PER_SIMD_CYCLE,
                                     "unit": "VALU instructions per SIMD-cycle", "frac": ipc / VALU_PEAK_PER_SIMD_CYCLE,
                                     "active_lane_fraction": counters["active_lane_fraction"],
                                     "valu_insts_per_launch": counters["valu_insts_per_launch"],
                                     "source": counters["source"], "measured_at": counters["measured_at"], "stale": counters["stale"]}
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
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(W, H, B, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
