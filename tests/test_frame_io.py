"""-m "not gpu": the host side of frame assembly and of the image writers."""
import numpy as np
import pytest


@pytest.mark.parametrize("W,H,world,rb", [(24, 64, 1, 8), (24, 64, 2, 8), (40, 1080, 8, 8), (7, 37, 4, 8), (16, 100, 3, 16), (5, 5, 8, 8)])
def test_deinterleave_index_math_matches_tilemap(api, W, H, world, rb):
    """pt_debug_gather_index is the host statement of the de-interleave kernel's map (frame[gid] =
    gathered[src[gid]]).  It must be the inverse of the scatter index distributed.TileMap builds for
    torch's index_copy_ (the exchange round 1 used), for equal-size padded slabs."""
    from opencl_path_tracer_amd.distributed import TileMap
    tm = TileMap(W, H, world, rb)
    src = api.gather_index(W, H, world, rb, tm.max_count)
    assert src.shape == (W * H,) and len(set(src.tolist())) == W * H
    dst = tm.gather_index("cpu").numpy()                  # gathered position -> frame row (W*H = padding)
    assert np.array_equal(dst[src], np.arange(W * H))
    for r in range(world):                                # and rank by rank, against the C ABI's own tiling
        sc = api.Scene(W, H, device=None, rank=r, world=world, rows_per_block=rb)
        assert sc.slab_pixels == tm.max_count
        ids = sc.local_pixel_ids()
        assert np.array_equal(src[ids], r * tm.max_count + np.arange(ids.size))


def _read_pfm(path):
    raw = open(path, "rb").read()
    head, rest = raw.split(b"\n", 3)[:3], raw.split(b"\n", 3)[3]
    assert head[0] == b"PF" and float(head[2]) < 0              # colour, little-endian
    W, H = [int(x) for x in head[1].split()]
    return np.frombuffer(rest, dtype="<f4").reshape(H, W, 3)


def _read_ppm(path):
    raw = open(path, "rb").read()
    head, rest = raw.split(b"\n", 3)[:3], raw.split(b"\n", 3)[3]
    assert head[0] == b"P6" and head[2] == b"255"
    W, H = [int(x) for x in head[1].split()]
    return np.frombuffer(rest, dtype=np.uint8).reshape(H, W, 3)


def test_image_writers_round_trip(api, tmp_path):
    W, H = 13, 7
    rng = np.random.RandomState(2)
    img = np.zeros((H * W, 4), dtype=np.float32)
    img[:, :3] = rng.uniform(-0.2, 1.4, (H * W, 3)).astype(np.float32)
    img[5, :3] = np.nan                                            # a black pixel's tone-mapped value (prog.cl:265-267)
    img[6, 0] = np.inf
    img[:, 3] = 1.0
    pfm, ppm = str(tmp_path / "a.pfm"), str(tmp_path / "a.ppm")
    api.write_pfm(pfm, img, W, H)
    api.write_ppm(ppm, img, W, H)
    got = _read_pfm(pfm)                                           # PFM rows are bottom-to-top = the buffer's own order
    assert np.array_equal(got.view(np.uint32), img[:, :3].reshape(H, W, 3).view(np.uint32))
    exp = img[:, :3].reshape(H, W, 3).copy()
    exp[~(exp > 0)] = 0.0                                          # NaN and negatives -> 0
    exp = np.minimum(exp, 1.0)
    exp8 = np.rint(exp * 255.0).astype(np.uint8)[::-1]             # PPM rows are top-to-bottom
    assert np.array_equal(_read_ppm(ppm), exp8)
    with pytest.raises(api.PtError) as e:
        api.write_ppm(str(tmp_path / "no_such_dir" / "x.ppm"), img, W, H)
    assert e.value.code == api.PT_EIO


def test_frame_calls_need_a_device(api, cb_spec):
    sc = api.Scene(16, 16, device=None, rank=0, world=2).load(cb_spec)
    for call in (sc.gather_frame, sc.read_frame, lambda: sc.write_pfm("/tmp/x.pfm"), lambda: sc.comm_init(bytes(128))):
        with pytest.raises(api.PtError) as e:
            call()
        assert e.value.code == api.PT_ENODEVICE
