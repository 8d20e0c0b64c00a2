"""-m "not gpu": the N > 1 path (tiling + all-gather + de-interleave, opencl_path_tracer_amd/
distributed.py) with world_size 2 and 3 over gloo on the CPU.  Each rank's slab is rendered by
the oracle here (no GPU in this container); on the GPU box bench.py feeds the same exchange with
the HIP-rendered slab.  The assembled frame must equal the single-process frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, rb, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as O
    from opencl_path_tracer_amd import api, scenes
    from opencl_path_tracer_amd.distributed import TileMap, exchange_frame
    spec = scenes.cornell_box(segments=8, rings=4)
    osc = O.load_scene(spec)
    cam = O.make_camera(spec.fov, spec.yaw, spec.pitch, spec.shift, W, H)
    fr = O.OracleFrame(W, H)
    fr.render(osc, cam, 3, 0, 2, nthreads=2)               # stand-in for this rank's HIP render
    tmap = TileMap(W, H, world, rb)
    # the C ABI's own view of the tiling (host-only context) must agree with TileMap
    ctx = api.Scene(W, H, device=None, rank=rank, world=world, rows_per_block=rb)
    assert np.array_equal(ctx.local_pixel_ids().astype(np.int64), tmap.ids(rank))
    slab = torch.zeros((tmap.max_count, 4), dtype=torch.float32)
    slab[: tmap.count(rank)] = torch.from_numpy(fr.colors()[tmap.ids(rank)].copy())
    frame = exchange_frame(slab, tmap)
    np.save(os.path.join(out_dir, "frame_%d.npy" % rank), frame.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "full.npy"), fr.colors().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,H,rb", [(2, 40, 8), (3, 37, 8), (2, 24, 16)])
def test_exchange_over_gloo(tmp_path, world, H, rb):
    W = 24
    mp.spawn(_worker, args=(world, _free_port(), W, H, rb, str(tmp_path)), nprocs=world, join=True)
    full = np.load(tmp_path / "full.npy")
    for r in range(world):
        got = np.load(tmp_path / ("frame_%d.npy" % r))
        assert got.shape == full.shape
        assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_tilemap_matches_reference_partition():
    sys.path.insert(0, ROOT)
    from opencl_path_tracer_amd.distributed import TileMap
    t = TileMap(1920, 1080, 8, 8)
    assert sum(t.counts) == 1920 * 1080 and t.max_count == 17 * 8 * 1920
    allids = np.concatenate([t.ids(r) for r in range(8)])
    assert np.array_equal(np.sort(allids), np.arange(1920 * 1080))
    idx = t.gather_index("cpu").numpy()
    assert (idx[idx < 1920 * 1080].size == 1920 * 1080) and idx.max() == 1920 * 1080


def test_band_context_is_a_one_rank_view_of_the_frame(tmp_path):
    """bench.py's self-validation (`frame_matches_n1`): a band of 24 rows as the tiling arguments of a ONE-rank context --
    the C ABI's own pixel map for those arguments must be exactly the band's global pixel ids, for frames whose height is
    and is not a multiple of the band."""
    sys.path.insert(0, ROOT)
    from opencl_path_tracer_amd import api
    from opencl_path_tracer_amd.distributed import band_context, band_pixel_ids, frame_matches_band
    for W, H in ((48, 1080), (40, 100), (16, 24), (8, 5)):
        band = band_context(H, 24)
        ids = band_pixel_ids(W, H, band)
        ctx = api.Scene(W, H, device=None, **band)
        assert np.array_equal(ctx.local_pixel_ids().astype(np.int64), ids)
        r0 = band["rank"] * band["rows_per_block"]
        assert ids[0] == r0 * W and ids.size == W * min(band["rows_per_block"], H - r0) and ids.size > 0
    # the comparison itself: bits, not values (a -0.0 is not a +0.0), RGB only
    frame = np.random.RandomState(1).rand(40 * 100, 4).astype(np.float32)
    band = band_context(100, 24)
    ids = band_pixel_ids(40, 100, band)
    good = frame[ids].copy()
    good[:, 3] = 7.0                                     # the pad lane is not part of the contract
    assert frame_matches_band(frame, good, ids)
    bad = good.copy()
    bad[5, 1] = np.nextafter(bad[5, 1], np.float32(2.0))
    assert not frame_matches_band(frame, bad, ids)
    frame[ids[3], 0] = 0.0
    neg = frame[ids].copy()
    neg[3, 0] = -0.0
    assert not frame_matches_band(frame, neg, ids)


def test_assembled_frame_matches_a_one_rank_band(tmp_path):
    """The N-rank path end to end over gloo with the check bench.py prints as `frame_matches_n1`: every rank's assembled frame
    passes the band comparison against a one-rank render of the band's rows (the oracle here), and a frame with one wrong
    row fails it."""
    world, W, H, rb = 2, 24, 40, 8
    mp.spawn(_worker, args=(world, _free_port(), W, H, rb, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from opencl_path_tracer_amd.distributed import band_context, band_pixel_ids, frame_matches_band
    full = np.load(tmp_path / "full.npy")
    band = band_context(H, 24)
    ids = band_pixel_ids(W, H, band)
    one_rank = full[ids]
    for r in range(world):
        got = np.load(tmp_path / ("frame_%d.npy" % r))
        assert frame_matches_band(got, one_rank, ids)
        got[ids[W * 9 + 2], 2] += 1.0
        assert not frame_matches_band(got, one_rank, ids)


def test_watchdog_turns_a_missing_rank_into_an_exit_code():
    """A phase that waits for a rank that never comes must end the process with a non-zero code (bench.py wraps every
    rendezvous / collective / synchronize in this), and a phase that finishes in time must not be disturbed."""
    import subprocess
    code = ("import sys, time; sys.path.insert(0, %r)\n"
            "from opencl_path_tracer_amd.distributed import Watchdog\n"
            "with Watchdog(5.0, 'quick phase', 1):\n    time.sleep(0.05)\n"
            "with Watchdog(0.3, 'all-gather with a missing rank', 1):\n    time.sleep(30)\n"
            "print('not reached')\n" % ROOT)
    t0 = __import__("time").time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "not reached" not in r.stdout and "all-gather with a missing rank" in r.stderr
    assert __import__("time").time() - t0 < 25
