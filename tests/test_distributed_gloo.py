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
