"""What pins the oracle (oracle/pt_oracle.c).

The reference has no tests, fixtures or golden outputs and cannot be built in this image, so the
oracle is "parity unpinned" against an execution of the reference.  These tests pin everything
that CAN be pinned: the RNG against libstdc++'s std::minstd_rand0 (the generator main.cpp:45
uses) and the C++ standard's known answers, the OpenCL struct layouts, the spec math against
libm within the OpenCL accuracy bounds, closed-form radiance cases read off prog.cl, and the
oracle's internal consistency (heap array == pointer tree == brute force).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest


def test_struct_layouts(oracle):
    L = oracle.lib()
    got = [L.orc_layout(i) for i in range(22)]
    # prog.cl:1-35 with float3 = 16 B (OpenCL 1.2 6.1.5): sizes and offsets, SURVEY 8(a) T1-T6
    want = [80, 16, 32, 48, 64, 68, 72,      # Material: size, ks, emission, F0, n, shininess, type
            32, 16,                          # Ray: size, D
            80, 48, 64,                      # Triangle: size, N, mati
            48, 16,                          # Node: size, bbox
            80, 64, 68,                      # Camera: size, XM, YM
            144, 16, 48, 64,                 # Hit: size, P, mati, mat
            32]                              # BBox
    assert got == want


def test_seed_sequence_known_answers(oracle):
    s = oracle.seed_sequence(10000)
    # first outputs of a default-constructed std::minstd_rand0 ...
    assert s[:5].tolist() == [16807, 282475249, 1622650073, 984943658, 1144108930]
    # ... and the C++ standard's check value [rand.predef]: 10000th invocation = 1043618065
    assert int(s[-1]) == 1043618065


def test_seed_sequence_matches_libstdcxx(oracle, tmp_path):
    """main.cpp:45,522-527 draws RNDS[i] from std::minstd_rand0: compare with the real thing."""
    src = tmp_path / "seeds.cpp"
    src.write_text("#include <random>\n#include <cstdio>\nint main(){std::minstd_rand0 g; for(int i=0;i<65536;++i) std::printf(\"%d\\n\", (int)g()); }\n")
    exe = tmp_path / "seeds"
    subprocess.run(["g++", "-O1", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert np.array_equal(np.array(out, dtype=np.int64), oracle.seed_sequence(65536).astype(np.int64))


def test_device_lcg_known_answer(oracle):
    """prog.cl:72-77 is std::minstd_rand (a = 48271): 10000th value from seed 1 is 399268537."""
    L = oracle.lib()
    seed = C.c_int(1)
    f = 0.0
    for _ in range(10000):
        f = L.orc_rand(C.byref(seed))
    assert seed.value == 399268537
    assert f == np.float32(np.float32(399268537) / np.float32(2147483648.0))
    # result range (0, 1]: n = 2^31-2 rounds up to 2^31 -> exactly 1.0
    seed = C.c_int(pow(48271, -1, 2147483647) * 2147483646 % 2147483647)
    assert L.orc_rand(C.byref(seed)) == 1.0 and seed.value == 2147483646


def _ulp_err(got, exact):
    got = np.asarray(got, dtype=np.float64)
    exact = np.asarray(exact, dtype=np.float64)
    ulp = np.spacing(np.abs(exact).astype(np.float32)).astype(np.float64)
    return np.abs(got - exact) / ulp


def test_spec_sincos_within_opencl_bounds(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(7)
    th = np.concatenate([rng.rand(20000) * 2 * math.pi, [0.0, 2 * math.pi, math.pi / 2, math.pi, 1e-8]]).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    gs, gc = [], []
    for t in th:
        L.orc_spec_sincosf(float(t), C.byref(s), C.byref(c))
        gs.append(s.value)
        gc.append(c.value)
    es, ec = np.sin(th.astype(np.float64)), np.cos(th.astype(np.float64))
    # OpenCL 1.2 table 7.4: sin, cos <= 4 ulp.  Near zeros of sin/cos the ulp of the result is
    # tiny, so bound the absolute error there instead (as the spec's own tests do).
    ok_s = (_ulp_err(gs, es) <= 1.0) | (np.abs(np.array(gs) - es) < 1e-9)
    ok_c = (_ulp_err(gc, ec) <= 1.0) | (np.abs(np.array(gc) - ec) < 1e-9)
    assert ok_s.all() and ok_c.all()


def test_spec_pow_within_opencl_bounds(oracle):
    L = oracle.lib()
    rng = np.random.RandomState(11)
    xs = np.concatenate([rng.rand(4000), [1.0, 0.5, 1e-3, 0.999999]]).astype(np.float32)
    for y in (0.4167, 1.0, 2.0, 5.0, 50.0, 200.0):
        yf = np.float32(y)
        got = np.array([L.orc_spec_powf(float(x), float(yf)) for x in xs])
        exact = np.power(xs.astype(np.float64), np.float64(yf))
        big = exact > 2.0 ** -120
        assert (_ulp_err(got[big], exact[big]) <= 1.0).all()          # OpenCL allows 16 ulp
        assert (got[~big] <= 2.0 ** -119).all()
    assert L.orc_spec_powf(0.0, 50.0) == 0.0
    assert L.orc_spec_powf(0.3, 0.0) == 1.0 and L.orc_spec_powf(0.0, 0.0) == 1.0
    x = np.float32(0.3)
    assert L.orc_spec_pow5(float(x)) == np.float32(np.float32(np.float32(x * x) * np.float32(x * x)) * x)


def test_material_constructor(oracle):
    # main.cpp:101-111 on the CHROMIUM row (main.cpp:760), evaluated independently in numpy f32
    N = np.array([3.10, 3.05, 2.05], dtype=np.float32)
    K = np.array([3.3, 3.3, 2.9], dtype=np.float32)
    m = oracle.make_material((0, 0, 0), (0, 0, 0), (0, 0, 0), N, K, 0.0, 1)[0]
    one = np.float32(1)
    a = (N - one) * (N - one)
    b = (N + one) * (N + one)
    assert np.array_equal(m["F0"][:3], (K * K + a) / (K * K + b))
    assert m["n"] == np.float32(np.float32(np.float32(N[0] + N[1]) + N[2]) / np.float32(3.0))
    # N = K = 0 gives F0 = 1
    m0 = oracle.make_material((0.3, 0.3, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)[0]
    assert np.array_equal(m0["F0"][:3], np.ones(3, np.float32)) and m0["type"] == 0


def test_camera_canonical_view(oracle):
    # fov 60, yaw 0, pitch 0: eye (500,500,-1299.037842), ahead = (W/2)/tan(30 deg) along +z
    c = oracle.make_camera(60.0, 0.0, 0.0, (0, 0, 0), 256, 256)[0]
    assert np.array_equal(c["eye"][:3], np.array([500.0, 500.0, -1299.037842], np.float32))
    assert np.array_equal(c["up"][:3], np.array([0, 128, 0], np.float32))
    assert np.array_equal(c["right"][:3], np.array([128, 0, 0], np.float32))
    ahead = c["lookat"][:3] - c["eye"][:3]
    assert ahead[0] == 0 and ahead[1] == 0 and abs(ahead[2] - 128 / math.tan(math.radians(30))) < 1e-3
    assert c["XM"] == 256 and c["YM"] == 256


def test_camera_ray_row0_is_bottom(oracle):
    cam = oracle.make_camera(60.0, 0.0, 0.0, (0, 0, 0), 64, 64)
    r = np.zeros(1, dtype=oracle.RAY)
    oracle.lib().orc_camera_get_ray(r.ctypes.data, 0, cam.ctypes.data, C.c_float(0.5), C.c_float(0.5))
    assert r[0]["D"][0] < 0 and r[0]["D"][1] < 0 and r[0]["D"][2] > 0          # id 0 looks down-left
    oracle.lib().orc_camera_get_ray(r.ctypes.data, 64 * 64 - 1, cam.ctypes.data, C.c_float(0.5), C.c_float(0.5))
    assert r[0]["D"][0] > 0 and r[0]["D"][1] > 0
    assert abs(np.linalg.norm(r[0]["D"][:3].astype(np.float64)) - 1) < 1e-6


def test_triangle_intersect_cases(oracle):
    L = oracle.lib()
    tri = oracle.make_triangle((0, 0, 5), (1, 0, 5), (0, 1, 5), 3)
    assert np.allclose(tri[0]["N"][:3], [0, 0, 1])
    ray = np.zeros(1, dtype=oracle.RAY)
    hit = np.zeros(1, dtype=oracle.HIT)

    def shoot(P, D):
        ray[0]["P"][:3] = P
        ray[0]["D"][:3] = D
        L.orc_triangle_intersect(hit.ctypes.data, tri.ctypes.data, ray.ctypes.data)
        return float(hit[0]["t"]), int(hit[0]["mati"])

    assert shoot((0.25, 0.25, 0), (0, 0, 1)) == (5.0, 3)          # front hit
    assert shoot((0.25, 0.25, 10), (0, 0, -1)) == (5.0, 3)        # back face also hits (no culling)
    assert shoot((0.25, 0.25, 10), (0, 0, 1))[0] == -1.0          # behind the origin: t < 0
    assert shoot((2.0, 2.0, 0), (0, 0, 1))[0] == -1.0             # outside
    assert shoot((0.25, 0.25, 0), (1, 0, 0))[0] == -1.0           # parallel: inf/NaN -> miss
    assert shoot((0.0, 0.0, 0), (0, 0, 1))[0] == 5.0              # exactly on a vertex: edges are >= 0
    deg = oracle.make_triangle((0, 0, 5), (1, 0, 5), (2, 0, 5), 0)  # degenerate: NaN normal never hits
    L.orc_triangle_intersect(hit.ctypes.data, deg.ctypes.data, ray.ctypes.data)
    assert hit[0]["t"] == -1.0


def test_bbox_slab_cases(oracle):
    L = oracle.lib()
    box = np.zeros(1, dtype=oracle.BBOX)
    box[0]["bl"][:3] = (0, 0, 0)
    box[0]["tr"][:3] = (1, 1, 1)
    ray = np.zeros(1, dtype=oracle.RAY)
    tmin, tmax = C.c_float(), C.c_float()

    def shoot(P, D):
        ray[0]["P"][:3] = P
        ray[0]["D"][:3] = D
        ok = L.orc_bbox_intersection(box.ctypes.data, ray.ctypes.data, C.byref(tmin), C.byref(tmax))
        return ok, tmin.value, tmax.value

    assert shoot((0.5, 0.5, -1), (0, 0, 1)) == (1, 1.0, 2.0)       # D.x = D.y = 0: divisions by zero give +-inf
    assert shoot((1.5, 0.5, -1), (0, 0, 1))[0] == 0                # misses in x
    ok, a, b = shoot((0.5, 0.5, 0.5), (0, 0, 1))                   # origin inside: tmin < 0 <= tmax
    assert ok == 1 and a == -0.5 and b == 0.5
    assert shoot((0.5, 0.5, 3), (0, 0, 1))[2] < 0                  # box behind: tmax < 0 (prog.cl:161 skips it)


def _single_quad_scene(oracle, mat_args, y=10.0):
    sc = oracle.OracleScene()
    sc.add_Material(*mat_args)
    e = 1.0e6      # large enough that every upward camera ray hits it
    sc.add_Triangle((-e, y, -e), (-e, y, e), (e, y, e), 0)
    sc.add_Triangle((e, y, e), (e, y, -e), (-e, y, -e), 0)
    sc.end_Obj()
    return sc


def test_analytic_direct_emitter(oracle):
    """A camera ray that hits an emitter first: color = emission*(1+1)*1*1*max(0,-D.N)
    (prog.cl:358-366 with all factors still 1), then the path leaves through a diffuse bounce and
    misses.  One sample, so colors == that value exactly (running mean of prog.cl:379 at s=0)."""
    em = (3.0, 2.0, 1.0)
    sc = _single_quad_scene(oracle, ((0, 0, 0), (0, 0, 0), em, (0, 0, 0), (0, 0, 0), 0.0, 3), y=2000.0)
    W = H = 8
    cam = oracle.make_camera(60.0, 0.0, -60.0, (0, 0, 0), W, H)     # pitch up towards the quad
    fr = oracle.OracleFrame(W, H)
    fr.generate_rays(cam)
    rays = fr.rays().copy()
    fr.trace_rays(sc, cam, 4, 0)
    cols = fr.colors()
    D = rays["D"][:, :3].astype(np.float64)
    cos = np.abs(D[:, 1])                                            # quad normal is +-y
    hitmask = D[:, 1] > 0
    assert hitmask.any()
    want = 2.0 * np.array(em)[None, :] * cos[:, None]
    assert np.allclose(cols[hitmask, :3], want[hitmask], rtol=2e-6)
    assert np.all(cols[~hitmask, :3] == 0)                           # black environment, prog.cl:367-376


def test_analytic_preview_mode(oracle):
    """iterations == 1: color = kd + emission of the first hit (prog.cl:323-325)."""
    kd = (0.25, 0.5, 0.75)
    sc = _single_quad_scene(oracle, (kd, (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0), y=2000.0)
    cam = oracle.make_camera(60.0, 0.0, -60.0, (0, 0, 0), 8, 8)
    fr = oracle.OracleFrame(8, 8)
    fr.render(sc, cam, 1, 0, 1)
    cols = fr.colors()[:, :3]
    hit = cols.sum(1) > 0
    assert hit.any() and np.array_equal(cols[hit], np.tile(np.array(kd, np.float32), (hit.sum(), 1)))


def test_draw_counts_per_material(oracle):
    """Draws per bounce: gen_ray 2, diffuse 2, mirror 0, glass 1, emitter 2, miss 0.
    Checked through the final LCG state of each pixel."""
    def advance(seed, k):
        for _ in range(k):
            seed = seed * 48271 % 2147483647
        return seed

    for mtype, per_hit in ((0, 2), (1, 0), (2, 1), (3, 2)):
        N = (1.5, 1.5, 1.5) if mtype == 2 else (0, 0, 0)
        sc = _single_quad_scene(oracle, ((0.3, 0.3, 0.3), (0, 0, 0), (1, 1, 1), N, (0, 0, 0), 50.0, mtype), y=2000.0)
        cam = oracle.make_camera(60.0, 0.0, -60.0, (0, 0, 0), 8, 8)
        fr = oracle.OracleFrame(8, 8)
        seeds0 = fr.rnds().copy()
        fr.generate_rays(cam)
        up = fr.rays()["D"][:, 1] > 0
        fr.trace_rays(sc, cam, 1, 0)          # one bounce: exactly one hit for the upward rays
        got = fr.rnds()
        for i in range(64):
            assert got[i] == advance(int(seeds0[i]), 2 + (per_hit if up[i] else 0))


def test_heap_pointer_brute_agree(oracle, cb_oracle_scene):
    """Reference traversal on the heap array (prog.cl:144-184) == same on the pointer tree ==
    exhaustive closest hit with first-met tie-break, on a full Cornell-box render."""
    W = H = 48
    cam = oracle.make_camera(60.0, 0.0, 0.0, (0, 0, 0), W, H)
    frames = []
    for mode in (0, 1, 2):
        fr = oracle.OracleFrame(W, H)
        fr.render(cb_oracle_scene, cam, 4, 0, 3, mode=mode, nthreads=8)
        frames.append((fr.colors().copy(), fr.rnds().copy()))
    for c, r in frames[1:]:
        assert np.array_equal(c.view(np.uint32), frames[0][0].view(np.uint32))
        assert np.array_equal(r, frames[0][1])


def test_reference_tree_shape(oracle, cb_oracle_scene):
    """NodeOnHost::build/convert (main.cpp:210-303): leaves hold <= 6 triangles, every triangle sits in
    exactly one leaf range, objects are appended with kd_tree_shift = size-1 (main.cpp:537-540)."""
    nodes = cb_oracle_scene.nodes()
    shifts = cb_oracle_scene.shifts()
    cnt = cb_oracle_scene.counts()
    assert cnt["ntris"] == 1932 and cnt["nobj"] == 3 and shifts[0] == 0
    leaves = nodes[nodes["trii"][:, 0] >= 0]
    sizes = leaves["trii"][:, 1] - leaves["trii"][:, 0]
    assert sizes.max() <= 6 and sizes.min() >= 1 and sizes.sum() == 1932
    covered = np.zeros(1932, dtype=int)
    for a, b in leaves["trii"]:
        covered[a:b] += 1
    assert (covered == 1).all()
    rank = cb_oracle_scene.encounter_rank()
    assert sorted(rank.tolist()) == list(range(1932))


def test_running_mean_and_sample_zero_reset(oracle, cb_oracle_scene):
    """prog.cl:312-314, 379: sample 0 overwrites colors; N samples in one call == N calls of one."""
    W = H = 16
    cam = oracle.make_camera(60.0, 0.0, 0.0, (0, 0, 0), W, H)
    a = oracle.OracleFrame(W, H)
    a.colors()[:] = 123.0
    a.render(cb_oracle_scene, cam, 4, 0, 5)
    b = oracle.OracleFrame(W, H)
    for s in range(5):
        b.generate_rays(cam)
        b.trace_rays(cb_oracle_scene, cam, 4, s)
    assert np.array_equal(a.colors().view(np.uint32), b.colors().view(np.uint32))
    assert np.array_equal(a.rnds(), b.rnds())


def test_tone_maps(oracle):
    L = oracle.lib()
    out = (C.c_float * 4)()
    L.orc_reinhard_tone_map(out, (C.c_float * 3)(0.0, 0.0, 0.0))
    assert math.isnan(out[0]) and out[3] == 1.0                    # black -> 0*0/0
    L.orc_reinhard_tone_map(out, (C.c_float * 3)(1.0, 1.0, 1.0))
    # L = 1 -> c * 0.5 -> sRGB(0.5) = 1.055*0.5^0.4167 - 0.055
    assert abs(out[0] - (1.055 * 0.5 ** 0.4167 - 0.055)) < 1e-6
    L.orc_filmic_tone_map(out, (C.c_float * 3)(0.0, 0.004, 1.0))
    assert out[0] == 0.0 and out[1] == 0.0
    c = 1.0 - 0.004
    assert abs(out[2] - (c * (c * 6.2 + 0.5)) / (c * (c * 6.2 + 1.7) + 0.06)) < 1e-6


def test_golden_regression(oracle, cb_oracle_scene):
    """tests/golden/cb_64x64_b4_s4.npz is the oracle's OWN output frozen at commit time (made by
    tests/golden/make_golden.py); it guards the oracle against accidental change.  It is not a
    reference output -- none exists."""
    path = os.path.join(os.path.dirname(__file__), "golden", "cb_64x64_b4_s4.npz")
    g = np.load(path)
    cam = oracle.make_camera(60.0, 0.0, 0.0, (0, 0, 0), 64, 64)
    fr = oracle.OracleFrame(64, 64)
    segs = fr.render(cb_oracle_scene, cam, 4, 0, 4, nthreads=8)
    assert np.array_equal(fr.colors()[:, :3].view(np.uint32), g["colors"].view(np.uint32))
    assert np.array_equal(fr.rnds(), g["rnds"])
    assert segs == int(g["segments"])


def test_analytic_mirror_then_emitter(oracle):
    """Camera -> mirror quad (type 1) -> emitter quad: color = emission * (1+1) * F * 1 * cos_e with
    F = F0 + (1-F0)(1-|N.D|)^5 (prog.cl:219-222, 341-345, 358-366) evaluated independently in numpy."""
    N = np.array([3.10, 3.05, 2.05], np.float32)
    K = np.array([3.3, 3.3, 2.9], np.float32)
    sc = oracle.OracleScene()
    sc.add_Material((0, 0, 0), (0, 0, 0), (0, 0, 0), N, K, 0.0, 1)           # mirror ceiling at y = 3000
    sc.add_Material((0, 0, 0), (0, 0, 0), (2.0, 3.0, 4.0), (0, 0, 0), (0, 0, 0), 0.0, 3)   # emitter floor at y = -3000
    e = 1.0e6
    sc.add_Triangle((-e, 3000, -e), (-e, 3000, e), (e, 3000, e), 0)
    sc.add_Triangle((e, 3000, e), (e, 3000, -e), (-e, 3000, -e), 0)
    sc.end_Obj()
    sc.add_Triangle((-e, -3000, -e), (-e, -3000, e), (e, -3000, e), 1)
    sc.add_Triangle((e, -3000, e), (e, -3000, -e), (-e, -3000, -e), 1)
    sc.end_Obj()
    W = H = 8
    cam = oracle.make_camera(60.0, 0.0, -60.0, (0, 0, 0), W, H)              # looking up at the mirror
    fr = oracle.OracleFrame(W, H)
    fr.generate_rays(cam)
    D = fr.rays()["D"][:, :3].astype(np.float64)
    fr.trace_rays(sc, cam, 2, 0)                                              # two segments: mirror, emitter
    cols = fr.colors()[:, :3].astype(np.float64)
    up = D[:, 1] > 0
    assert up.sum() > 32
    F0 = ((K.astype(np.float64) ** 2 + (N.astype(np.float64) - 1) ** 2) / (K.astype(np.float64) ** 2 + (N.astype(np.float64) + 1) ** 2))
    cosm = np.abs(D[:, 1])                      # |N.D| at the mirror; the reflected ray hits the floor with the same cosine
    F = F0[None, :] + (1 - F0[None, :]) * ((1 - cosm) ** 5)[:, None]
    want = np.array([2.0, 3.0, 4.0])[None, :] * 2.0 * F * cosm[:, None]
    assert np.allclose(cols[up], want[up], rtol=3e-6)


def test_analytic_diffuse_factor(oracle):
    """One diffuse bounce then the lamp: color = E * (fL + fB) * cos_lamp with fL = kd * max(0, N.D'),
    fB = ks * max(0, N.H)^shininess, H = normalize(view + D') (prog.cl:329-340, 358-366), where D' is the
    cosine-sampled direction the oracle drew.  Recomputed from the rays buffer after one trace_rays."""
    kd, ks, shin = (0.3, 0.2, 0.1), (0.3, 0.3, 0.3), 20.0
    sc = oracle.OracleScene()
    sc.add_Material(kd, ks, (0, 0, 0), (0, 0, 0), (0, 0, 0), shin, 0)         # diffuse floor far below
    sc.add_Material((0, 0, 0), (0, 0, 0), (5.0, 5.0, 5.0), (0, 0, 0), (0, 0, 0), 0.0, 3)   # lamp ceiling
    e = 1.0e6
    sc.add_Triangle((-e, -2000, -e), (-e, -2000, e), (e, -2000, e), 0)
    sc.add_Triangle((e, -2000, e), (e, -2000, -e), (-e, -2000, -e), 0)
    sc.end_Obj()
    sc.add_Triangle((-e, 4000, -e), (-e, 4000, e), (e, 4000, e), 1)
    sc.add_Triangle((e, 4000, e), (e, 4000, -e), (-e, 4000, -e), 1)
    sc.end_Obj()
    W = H = 8
    cam = oracle.make_camera(60.0, 0.0, 60.0, (0, 0, 0), W, H)                # looking down at the floor
    fr = oracle.OracleFrame(W, H)
    fr.generate_rays(cam)
    D0 = fr.rays()["D"][:, :3].astype(np.float64)
    eye = cam[0]["eye"][:3].astype(np.float64)
    fr2 = oracle.OracleFrame(W, H)                                            # same seeds: 1 segment only -> rays = bounce ray
    fr2.generate_rays(cam)
    fr2.trace_rays(sc, cam, 1, 0)
    D1 = fr2.rays()["D"][:, :3].astype(np.float64)
    P1 = fr2.rays()["P"][:, :3].astype(np.float64)
    fr.trace_rays(sc, cam, 2, 0)
    cols = fr.colors()[:, :3].astype(np.float64)
    down = D0[:, 1] < 0
    assert down.sum() > 32
    Nf = np.array([0.0, 1.0, 0.0])
    idiff = np.maximum(0, D1 @ Nf)
    view = eye[None, :] - P1                                                  # hit point ~ P1 (offset 1e-3 along N)
    view /= np.linalg.norm(view, axis=1)[:, None]
    Hh = view + D1
    Hh /= np.linalg.norm(Hh, axis=1)[:, None]
    ispec = np.maximum(0, Hh @ Nf)
    fL = np.array(kd)[None, :] * idiff[:, None]
    fB = np.array(ks)[None, :] * (ispec ** shin)[:, None]
    coslamp = np.abs(D1[:, 1])
    want = 5.0 * (fL + fB) * coslamp[:, None]
    assert np.allclose(cols[down], want[down], rtol=2e-5)


def test_analytic_dielectric_split(oracle):
    """Camera -> glass plane (type 2, n = 1.5) -> emitter above (refracted path) or below (reflected
    path): color = E * 2 * fR * cos_e with fR = (1-F)/(1-prob) on refraction, F/prob on reflection,
    prob = mean(F) (prog.cl:228-245, 346-357); the refracted direction is D/n + N(cosa/n - sqrt(disc))."""
    n = 1.5
    sc = oracle.OracleScene()
    sc.add_Material((0, 0, 0), (0, 0, 0), (0, 0, 0), (n, n, n), (0, 0, 0), 0.0, 2)
    sc.add_Material((0, 0, 0), (0, 0, 0), (1.0, 2.0, 3.0), (0, 0, 0), (0, 0, 0), 0.0, 3)
    e = 1.0e6
    sc.add_Triangle((-e, 2000, -e), (-e, 2000, e), (e, 2000, e), 0)          # glass
    sc.add_Triangle((e, 2000, e), (e, 2000, -e), (-e, 2000, -e), 0)
    sc.end_Obj()
    for y in (6000.0, -4000.0):                                               # emitters above and below
        sc.add_Triangle((-e, y, -e), (-e, y, e), (e, y, e), 1)
        sc.add_Triangle((e, y, e), (e, y, -e), (-e, y, -e), 1)
        sc.end_Obj()
    W = H = 16
    cam = oracle.make_camera(50.0, 0.0, -55.0, (0, 0, 0), W, H)
    a = oracle.OracleFrame(W, H)
    a.generate_rays(cam)
    D0 = a.rays()["D"][:, :3].astype(np.float64)
    b = oracle.OracleFrame(W, H)
    b.generate_rays(cam)
    b.trace_rays(sc, cam, 1, 0)                                               # after the glass interaction only
    D1 = b.rays()["D"][:, :3].astype(np.float64)
    a.trace_rays(sc, cam, 2, 0)
    cols = a.colors()[:, :3].astype(np.float64)
    up = D0[:, 1] > 0.05
    assert up.sum() > 200
    F0 = ((n - 1) ** 2) / ((n + 1) ** 2)
    cosa = np.abs(D0[:, 1])                       # N flipped against the ray: cosa = -D.N = |D.y|
    F = F0 + (1 - F0) * (1 - cosa) ** 5
    prob = F                                      # all three channels equal
    refracted = D1[:, 1] > 0
    assert refracted[up].any() and (~refracted[up]).any()          # both outcomes occur among 200+ pixels
    fR = np.where(refracted, (1 - F) / (1 - prob), F / prob)
    disc = 1 - (1 - cosa ** 2) / n / n
    # refracted direction (flipped normal is -y): D/n + N*(cosa/n - sqrt(disc)), then normalised
    Dr = D0 / n + np.array([0.0, -1.0, 0.0])[None, :] * (cosa / n - np.sqrt(disc))[:, None]
    Dr /= np.linalg.norm(Dr, axis=1)[:, None]
    assert np.allclose(D1[up & refracted], Dr[up & refracted], atol=2e-6)
    cos_e = np.abs(D1[:, 1])
    want = np.array([1.0, 2.0, 3.0])[None, :] * 2.0 * (fR * cos_e)[:, None]
    assert np.allclose(cols[up], want[up], rtol=5e-6)


@pytest.mark.parametrize("which", ["walls", "glossy", "cornell", "mesh"])
def test_oracle_against_the_float64_replay_model(oracle, api, which):
    """The oracle itself against the independent float64 numpy model of the whole hot path that the GPU tests use
    (tests/test_gpu_closed_form.py::replay_model, written from prog.cl's text: exact triangle test over all triangles, closest
    hit, diffuse / mirror / glass / emitter, the LCG stream per pixel): the Cornell walls at five bounces and BASELINE's Cornell
    box at eight.  Same criteria as on the GPU: colours within 2e-4 and the SAME final LCG state in every pixel whose
    decisions were clear of float32 / float64 ties -- the state encodes the sequence of materials each path met."""
    import importlib.util
    import os
    from opencl_path_tracer_amd import scenes
    here = os.path.dirname(os.path.abspath(__file__))
    spec_ = importlib.util.spec_from_file_location("closed_form_model", os.path.join(here, "test_gpu_closed_form.py"))
    cf = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(cf)
    m = cf.replay_model(api, which)
    spec = scenes.SceneSpec(materials=list(scenes.BUILTIN_MATERIALS), name=which)
    spec.objects = list(m["objects"])
    osc = oracle.load_scene(spec)
    cam = oracle.make_camera(spec.fov, spec.yaw, spec.pitch, spec.shift, m["W"], m["H"])
    fr = oracle.OracleFrame(m["W"], m["H"])
    fr.render(osc, cam, m["ITER"], 0, m["S"], mode=0, nthreads=8)
    got = fr.colors()[:, :3].astype(np.float64)
    safe = m["safe"]
    assert safe.sum() > 0.75 * safe.size
    rtol = 2e-3 if which == "glossy" else 2e-4                    # pow(., 200) multiplies the float32 error of its argument by 200
    assert np.allclose(got[safe], m["exp"][safe], rtol=rtol, atol=1e-4), float(np.abs(got[safe] - m["exp"][safe]).max())
    assert np.array_equal(fr.rnds().astype(np.int64)[safe], m["state"][safe])
