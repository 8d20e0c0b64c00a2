"""-m gpu: closed-form radiance cases asserted directly on read_colors() -- NO oracle involved.

The expected values are derived here from prog.cl's text alone, with an independent float64 numpy model of
the two things they depend on: the per-pixel LCG stream (prog.cl:72-77 + the host seeding main.cpp:522-527,
exact integer arithmetic) and the pinhole camera (prog.cl:82-92, main.cpp:311-347).  They pin the device code
against a misreading that the oracle (written by the same hand) could share."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M31 = 2147483647
EYE_Z = -1299.037842
LAMP_E = np.array([120.0, 100.0, 80.0])        # main.cpp:753
SUN_E = np.array([300.0, 250.0, 200.0])        # main.cpp:754


def seeds(n):                                   # std::minstd_rand0, default seed, pixel order (main.cpp:45, 522-527)
    out = np.empty(n, dtype=np.int64)
    x = 1
    for i in range(n):
        x = (x * 16807) % M31
        out[i] = x
    return out


def draw(state):                                # prog.cl:72-77: returns (new state, float value n / 2^31)
    state = (state * 48271) % M31
    return state, state.astype(np.float64) / 2147483648.0


def camera_dz(W, H, fov_deg, ids, r1, r2):
    """z component of the unit camera-ray direction for yaw = pitch = 0 (prog.cl:82-92)."""
    x = (ids % W) + r1
    y = (ids // W) + r2
    ahead = (W / 2.0) / np.tan(np.float32(fov_deg / 2.0 / 180.0 * np.float32(3.141593)).astype(np.float64))
    dx = (W / 2.0) * (2.0 * x / W - 1.0)
    dy = (H / 2.0) * (2.0 * y / H - 1.0)
    return ahead / np.sqrt(dx * dx + dy * dy + ahead * ahead)


def quad(z, half=1.0e5):
    a, b, c, d = (-half, -half, z), (half, -half, z), (half, half, z), (-half, half, z)
    return np.array([[a, b, c], [a, c, d]], dtype=np.float32)


def big_tri(z, L=1.0e7):
    """One huge triangle around the view axis.  (A +-1e7 quad leaks: a point within ~1 unit of the shared
    diagonal fails BOTH triangles' inside tests -- the edge function there is below the rounding noise of its
    2e14-sized terms -- and the view axis (500, 500) lies on that diagonal.  Reference behaviour, reproduced
    bit for bit by the device; it just is not what this test is about.)"""
    return np.array([[(-L, -L, z), (3 * L, -L, z), (-L, 3 * L, z)]], dtype=np.float32)


def build(api, W, H, fov, objects):
    from opencl_path_tracer_amd import scenes
    sc = api.Scene(W, H)
    for m in scenes.BUILTIN_MATERIALS:
        sc.add_Material(*m)
    for verts, mat in objects:
        sc.add_Triangles(api.triangles_from_vertices(verts, np.full(verts.shape[0], mat, dtype=np.uint16)))
        sc.end_Obj()
    sc.upload_Triangles()
    sc.upload_Materials()
    sc.set_view(fov, 0.0, 0.0, (0.0, 0.0, 0.0))
    return sc


@pytest.mark.parametrize("variant", [0, 1])
def test_emitter_seen_in_a_mirror_at_normal_incidence(api, variant):
    """Camera -> chromium mirror (type 1, prog.cl:341-345) -> emitter behind the camera (type 3,
    prog.cl:358-366), iterations = 2.  factor_S = Fresnel = F0 + (1 - F0)(1 - |N.D|)^5 = F0 to float precision
    at (near) normal incidence; the emitter adds emission x (factor_L + factor_B) x factor_S x |D.N| with
    factor_L = factor_B = 1: every sample is 2 E F0 cos(theta), cos(theta) = the camera ray's z component.
    Draws per sample: 2 (gen_ray) + 0 (mirror) + 2 (emitter's continuation ray) = 4."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    S, fov = 24, 2.0
    sc = build(api, W, H, fov, [(quad(1000.0), scenes.CHROMIUM), (quad(-3000.0), scenes.LAMP)])
    sc.set_option("variant", variant)
    sc.iterations = 2
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    n_, k_ = np.array([3.10, 3.05, 2.05], np.float32), np.array([3.3, 3.3, 2.9], np.float32)     # main.cpp:760
    F0 = ((k_ * k_ + (n_ - 1) * (n_ - 1)) / (k_ * k_ + (n_ + 1) * (n_ + 1))).astype(np.float64)  # main.cpp:105-109
    state = seeds(W * H)
    ids = np.arange(W * H)
    mean = np.zeros(W * H)
    for s in range(S):
        state, r1 = draw(state)
        state, r2 = draw(state)
        mean += camera_dz(W, H, fov, ids, r1, r2)
        state, _ = draw(state)
        state, _ = draw(state)
    mean /= S
    exp = 2.0 * LAMP_E[None, :] * F0[None, :] * mean[:, None]
    assert np.allclose(got, exp, rtol=2e-5, atol=0), float(np.abs(got / exp - 1).max())
    assert np.array_equal(sc.read_rnds().astype(np.int64), state)        # exactly 4 draws per sample


@pytest.mark.parametrize("variant", [0, 1])
def test_glass_slab_at_normal_incidence(api, variant):
    """Camera -> glass slab (two parallel type-2 faces, n = 1.5, prog.cl:228-245, 346-357), iterations = 3.
    F = F0 = 0.04 on both faces, prob = mean(F) = F0, so the split weights are exactly 1: F / prob = 1 when
    the path reflects (rnd <= prob), (1 - F) / (1 - prob) = 1 when it refracts.  Per sample:
      front face reflects            -> SUN emitter behind the camera:  2 E_sun cos,   draws 2 + 1 + 2 + 1
      front refracts, back refracts  -> LAMP emitter behind the slab:   2 E_lamp cos,  draws 2 + 1 + 1 + 2
      front refracts, back reflects  -> still inside after 3 segments:  0,             draws 2 + 1 + 1 + 1
    (cos = the camera ray's z component: mirror reflection and a parallel slab both preserve it; the slab's faces are
    1e7-sized triangles so that the continuation ray leaving the SUN emitter misses it with probability ~1e-7 per event.)  Which
    branch a sample takes is decided by the LCG stream, replayed here in exact integer arithmetic."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    S, fov = 64, 2.0
    sc = build(api, W, H, fov, [(big_tri(1000.0), scenes.GLASS), (big_tri(1100.0), scenes.GLASS), (quad(2000.0), scenes.LAMP), (quad(-3000.0), scenes.SUN)])
    sc.set_option("variant", variant)
    sc.iterations = 3
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    prob = float(np.float32(0.25) / np.float32(6.25))                    # F0 = ((n-1)/(n+1))^2, main.cpp:105-109
    thr = prob * 2147483648.0
    state = seeds(W * H)
    ids = np.arange(W * H)
    acc = np.zeros((W * H, 3))
    n_reflect = n_through = n_trapped = 0
    safe = np.ones(W * H, bool)
    for s in range(S):
        state, r1 = draw(state)
        state, r2 = draw(state)
        cz = camera_dz(W, H, fov, ids, r1, r2)
        state, r3 = draw(state)                                          # front face
        safe &= np.abs(state - thr) > 64
        reflect = r3 <= prob
        s4, r4 = draw(state)                                             # back face (refracted paths) / emitter ray (reflected paths)
        safe &= reflect | (np.abs(s4 - thr) > 64)
        through = ~reflect & (r4 > prob)
        trapped = ~reflect & ~through
        acc += np.where(reflect[:, None], 2.0 * SUN_E[None, :] * cz[:, None], 0.0) + np.where(through[:, None], 2.0 * LAMP_E[None, :] * cz[:, None], 0.0)
        s5, _ = draw(s4)
        s6, _ = draw(s5)
        state = np.where(trapped, s5, s6)                                # 5 draws when trapped, 6 otherwise
        n_reflect += int(reflect.sum())
        n_through += int(through.sum())
        n_trapped += int(trapped.sum())
    exp = acc / S
    assert n_reflect > 1500 and n_trapped > 1500 and n_through > 50000 and safe.sum() > W * H - 4
    assert abs(n_reflect / (W * H * S) - 0.04) < 0.004                   # the split frequency is Fresnel's
    ok = safe
    assert np.allclose(got[ok], exp[ok], rtol=1e-4, atol=1e-3), float(np.abs(got[ok] - exp[ok]).max())
    assert np.array_equal(sc.read_rnds().astype(np.int64)[ok], state[ok])


@pytest.mark.parametrize("variant", [0, 1])
def test_diffuse_floor_under_an_emitting_ceiling(api, variant):
    """Camera -> diffuse floor (type 0, kd = 0.3, ks = 0; prog.cl:329-340, 186-218) -> emitting ceiling
    (type 3), iterations = 2.  The new direction is cosine-distributed about the normal (pdf cos/pi) and the
    kernel weights it by kd max(0, N.D') ON TOP of that (the reference's estimator); floor and ceiling are
    parallel, so the emitter's cosine is the same angle: a sample is E kd cos^2(theta) and its expectation
    E kd * Int cos^2 (cos/pi) dw = E kd / 2.  (factor_B = ks pow(...) = 0.)  Statistical pin of the sampler and
    of the estimator: 524,288 samples, sigma of the frame mean = 0.08 %.  Exactly 6 draws per sample."""
    from opencl_path_tracer_amd import scenes
    W = H = 64
    S = 128
    floor = big_tri(0.0)[:, :, [0, 2, 1]].copy()            # the y = 0 plane
    ceiling = big_tri(1000.0)[:, :, [0, 2, 1]].copy()       # the y = 1000 plane
    sc = build(api, W, H, 30.0, [(floor, scenes.WHITE_DIFFUSE), (ceiling, scenes.LAMP)])
    sc.set_view(30.0, 0.0, 40.0, (0.0, 0.0, 0.0))           # pitched down: every pixel sees the floor
    sc.set_option("variant", variant)
    sc.iterations = 2
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    assert (got.sum(1) > 0).all()
    exp = LAMP_E * 0.3 / 2.0
    assert np.allclose(got.mean(0), exp, rtol=5 * 0.0008), got.mean(0) / exp
    per_pixel_sigma = 0.577 / np.sqrt(S)                      # sigma / mean of one pixel's 128-sample mean
    assert np.abs(got / exp[None, :] - 1.0).max() < 6 * per_pixel_sigma
    state = seeds(W * H)
    for _ in range(6 * S):
        state, _ = draw(state)
    assert np.array_equal(sc.read_rnds().astype(np.int64), state)


# ----------------------------------------------------------------------------------------------------------------
# Round 3: four more pins that never touch the oracle.  Same rule: float64 numpy models written from prog.cl /
# main.cpp text alone, asserted on read_colors() / read_rnds(), both kernel variants.

def draw32(state):
    """prog.cl:72-77 with the float conversion spelled out: (float)n rounds the 31-bit state to 24 bits BEFORE the
    division by 2147483647.0f (= 2^31 as a float)."""
    state = (state * 48271) % M31
    return state, state.astype(np.float32).astype(np.float64) / 2147483648.0


def camera_dir(W, H, fov_deg, ids, r1, r2):
    """Unit camera-ray direction for yaw = pitch = 0: right = +x, up = +y, ahead = +z (main.cpp:322-341, prog.cl:82-92)."""
    x = (ids % W) + r1
    y = (ids // W) + r2
    ahead = (W / 2.0) / np.tan(np.float32(fov_deg / 2.0 / 180.0 * np.float32(3.141593)).astype(np.float64))
    dx = (W / 2.0) * (2.0 * x / W - 1.0)
    dy = (H / 2.0) * (2.0 * y / H - 1.0)
    inv = 1.0 / np.sqrt(dx * dx + dy * dy + ahead * ahead)
    return np.stack([dx * inv, dy * inv, ahead * inv], axis=1)


def tri_around(center, u, v, size):
    """One triangle in the plane through `center` spanned by u, v, containing the disc of radius `size` around center."""
    c, u, v = (np.asarray(a, dtype=np.float64) for a in (center, u, v))
    pts = [c - 2 * size * u - 2 * size * v, c + 4 * size * u - 2 * size * v, c - 2 * size * u + 4 * size * v]
    return np.array([pts], dtype=np.float32)


@pytest.mark.parametrize("variant", [0, 1])
def test_glossy_lobe_with_halfway_vector_to_the_camera_eye(api, variant):
    """Camera -> wall facing it (type 0 with ks != 0: PURPLE_SPECULAR kd (.3, 0, 0), ks .3, shininess 200) ->
    emitter behind the camera, iterations = 2.  prog.cl:329-340: the continuation ray is cosine-sampled about the
    flipped normal N = (0, 0, -1); by prog.cl:186-218 with |N.z| > 1e-3: Z = (-N.z, 0, N.x) = (1, 0, 0),
    X = N x Z = (0, -1, 0), D' = X r cos(th) + N sqrt(1 - rnd1) + Z r sin(th).  factor_L = kd max(0, N.D'),
    factor_B = ks pow(max(0, N.H), shininess) with H = normalize(normalize(eye - hit.P) + D') -- the CAMERA EYE
    (prog.cl:78-80, 335-338).  The emitter's plane is parallel to the wall, so its cosine is N.D' again:
    sample = E (factor_L + factor_B) (N.D').  6 draws per sample.  pow(c, 200) multiplies the float error of c by
    200, hence rtol 5e-4 on the per-pixel mean."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    S, fov = 48, 2.0
    sc = build(api, W, H, fov, [(big_tri(1000.0), scenes.PURPLE_SPECULAR), (big_tri(-3000.0), scenes.LAMP)])
    sc.set_option("variant", variant)
    sc.iterations = 2
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    kd, ks, shin = np.array([0.3, 0.0, 0.0]), np.array([0.3, 0.3, 0.3]), 200.0            # main.cpp:758
    kd, ks = kd.astype(np.float32).astype(np.float64), ks.astype(np.float32).astype(np.float64)
    eye = np.array([500.0, 500.0, float(np.float32(EYE_Z))])
    state = seeds(W * H)
    ids = np.arange(W * H)
    acc = np.zeros((W * H, 3))
    lobe_share = 0.0
    for s in range(S):
        state, r1 = draw32(state)
        state, r2 = draw32(state)
        D = camera_dir(W, H, fov, ids, r1, r2)
        hp = eye[None, :] + D * ((1000.0 - eye[2]) / D[:, 2])[:, None]
        state, u1 = draw32(state)
        state, u2 = draw32(state)
        r, th, c = np.sqrt(u1), 2.0 * np.pi * u2, np.sqrt(1.0 - u1)
        Dn = np.stack([r * np.sin(th), -r * np.cos(th), -c], axis=1)                       # X x + N z + Z y
        view = eye[None, :] - hp
        view /= np.linalg.norm(view, axis=1)[:, None]
        Hv = view + Dn
        Hv /= np.linalg.norm(Hv, axis=1)[:, None]
        nh = np.maximum(0.0, -Hv[:, 2])
        fL = kd[None, :] * c[:, None]
        fB = ks[None, :] * (nh ** shin)[:, None]
        acc += LAMP_E[None, :] * (fL + fB) * c[:, None]
        lobe_share += float((fB[:, 1] * c).sum())
        state, _ = draw32(state)
        state, _ = draw32(state)
    exp = acc / S
    assert lobe_share / (W * H * S) > 0.001            # the lobe term is really exercised
    assert (exp[:, 1] > 0).all() and np.allclose(exp[:, 1] * LAMP_E[2], exp[:, 2] * LAMP_E[1])   # G, B are pure lobe
    assert np.allclose(got, exp, rtol=5e-4, atol=1e-6), float(np.abs(got / exp - 1).max())
    assert np.array_equal(sc.read_rnds().astype(np.int64), state)


@pytest.mark.parametrize("variant", [0, 1])
def test_total_internal_reflection_inside_a_prism(api, variant):
    """Camera -> leg A of a right-angle glass prism (z = 1000, normal incidence) -> hypotenuse at 45 degrees from
    INSIDE -> leg C (x = 700, normal incidence) -> SUN emitter beyond it, iterations = 4.  prog.cl:228-245: inside,
    n becomes 1/1.5, disc = 1 - (1 - cos^2 45) 1.5^2 = -0.125 < 0: the path takes the mirror branch whatever the
    draw says (the draw is consumed), `in` stays set and factor_R *= F (1 / prob) = 1 (prog.cl:351-356).  A kernel
    that forgot the 1/n flip (disc = 0.78) would refract 96 % of these paths out through the hypotenuse into the void.
    Per sample, decided by the LCG stream:
      A reflects  (r <= F0)                 -> void:                      0,               draws 2 + 1
      A refracts, TIR, C refracts (r > F0)  -> SUN:  2 E_sun cos,  draws 2 + 1 + 1 + 1 + 2
      A refracts, TIR, C reflects           -> hypotenuse again, out of iterations: 0,     draws 2 + 1 + 1 + 1 + 1
    The mirror at 45 degrees swaps the x and z components and the two refractions at normal-ish incidence undo
    each other, so the direction that reaches the emitter plane x = 3000 is (d.z, d.y, d.x): cos = the camera ray's z."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    S, fov = 64, 2.0
    s2 = np.sqrt(0.5)
    leg_a = tri_around((500.0, 500.0, 1000.0), (1, 0, 0), (0, 1, 0), 80.0)
    hyp = tri_around((500.0, 500.0, 1100.0), (s2, 0, s2), (0, 1, 0), 120.0)                # the plane z = x + 600
    leg_c = tri_around((700.0, 500.0, 1100.0), (0, 1, 0), (0, 0, 1), 80.0)
    # (the emitter is small on purpose: a path that leg A reflects travels back along -z with a slight +x drift and would
    # meet an unbounded x = 3000 plane a quarter of a million units away)
    sun = tri_around((3000.0, 500.0, 1100.0), (0, 1, 0), (0, 0, 1), 300.0)
    sc = build(api, W, H, fov, [(leg_a, scenes.GLASS), (hyp, scenes.GLASS), (leg_c, scenes.GLASS), (sun, scenes.SUN)])
    sc.set_option("variant", variant)
    sc.iterations = 4
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    prob = float(np.float32(0.25) / np.float32(6.25))
    thr = prob * 2147483648.0
    state = seeds(W * H)
    ids = np.arange(W * H)
    acc = np.zeros((W * H, 3))
    safe = np.ones(W * H, bool)
    n_out = n_back = n_refl = 0
    for s in range(S):
        state, r1 = draw32(state)
        state, r2 = draw32(state)
        cz = camera_dir(W, H, fov, ids, r1, r2)[:, 2]
        sA, rA = draw32(state)                                                             # leg A
        safe &= np.abs(sA - thr) > 64
        reflA = rA <= prob
        sH, _ = draw32(sA)                                                                 # hypotenuse: consumed, ignored
        sC, rC = draw32(sH)                                                                # leg C
        safe &= reflA | (np.abs(sC - thr) > 64)
        out = ~reflA & (rC > prob)
        back = ~reflA & ~out
        acc += np.where(out[:, None], 2.0 * SUN_E[None, :] * cz[:, None], 0.0)
        s6, _ = draw32(sC)
        s7, _ = draw32(s6)
        state = np.where(reflA, sA, np.where(out, s7, s6))
        n_out += int(out.sum()); n_back += int(back.sum()); n_refl += int(reflA.sum())
    exp = acc / S
    assert n_out > 55000 and n_back > 1500 and n_refl > 1500 and safe.sum() > W * H - 4
    ok = safe
    assert np.allclose(got[ok], exp[ok], rtol=1e-4, atol=1e-3), float(np.abs(got[ok] - exp[ok]).max())
    assert np.array_equal(sc.read_rnds().astype(np.int64)[ok], state[ok])


@pytest.mark.parametrize("variant", [0, 1])
def test_preview_mode_on_the_gpu(api, variant):
    """iterations == 1 (prog.cl:323-325): color = kd + emission at the first hit, THEN the material branch runs with
    both factors still 1: a diffuse pixel reads kd, a directly seen emitter reads emission + 2 emission cos
    (prog.cl:358-366), a mirror reads 0 + 0, a miss reads 0.  Left half of the view: RED_DIFFUSE wall, right half:
    LAMP (pixel column x + rnd < W/2 <=> world x < 500: right = +x, main.cpp:323); top rows: nothing."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    S, fov = 16, 2.0
    far = 1.0e5
    wall = np.array([[(500.0, -far, 1000.0), (-far, -far, 1000.0), (-far, 520.0, 1000.0)],
                     [(500.0, -far, 1000.0), (-far, 520.0, 1000.0), (500.0, 520.0, 1000.0)]], dtype=np.float32)
    lamp = np.array([[(500.0, -far, 1000.0), (far, -far, 1000.0), (far, 520.0, 1000.0)],
                     [(500.0, -far, 1000.0), (far, 520.0, 1000.0), (500.0, 520.0, 1000.0)]], dtype=np.float32)
    sc = build(api, W, H, fov, [(wall, scenes.RED_DIFFUSE), (lamp, scenes.LAMP)])
    sc.set_option("variant", variant)
    sc.iterations = 1
    sc.render(S)
    got = sc.read_colors()[:, :3].astype(np.float64)
    state = seeds(W * H)
    ids = np.arange(W * H)
    px, py = ids % W, ids // W
    eye_z = float(np.float32(EYE_Z))
    acc = np.zeros((W * H, 3))
    hits = np.zeros(W * H, bool)
    clean = np.ones(W * H, bool)
    kd_red = np.array([0.3, 0.1, 0.1], np.float32).astype(np.float64)
    for s in range(S):
        state, r1 = draw32(state)
        state, r2 = draw32(state)
        D = camera_dir(W, H, fov, ids, r1, r2)
        y_at_wall = 500.0 + D[:, 1] / D[:, 2] * (1000.0 - eye_z)
        hit = y_at_wall < 520.0
        clean &= np.abs(y_at_wall - 520.0) > 1e-2
        left = px < W // 2
        val = np.where(left[:, None], kd_red[None, :], LAMP_E[None, :] * (1.0 + 2.0 * D[:, 2])[:, None])
        acc += np.where(hit[:, None], val, 0.0)
        s3, _ = draw32(state)
        s4, _ = draw32(s3)
        state = np.where(hit, s4, state)                                  # diffuse and emitter both draw 2; a miss draws none
        hits |= hit
    exp = acc / S
    ok = clean & (px != W // 2 - 1) & (px != W // 2)
    assert hits.sum() > W * H // 2 and (~hits).sum() > W * 4
    assert np.allclose(got[ok], exp[ok], rtol=2e-5, atol=1e-7), float(np.abs(got[ok] - exp[ok]).max())
    assert np.array_equal(sc.read_rnds().astype(np.int64)[ok], state[ok])


@pytest.mark.parametrize("variant", [0, 1])
def test_running_mean_over_several_launches(api, variant):
    """prog.cl:379: colors = (colors * current_sample + color) / (current_sample + 1), current_sample carried by the
    host across launches (main.cpp:686).  Three render() calls of 5 + 11 + 8 samples must give the arithmetic mean of
    the 24 per-sample values of the float64 model (mirror at normal incidence: 2 E F0 cos), and the very bits of one
    render(24) call."""
    from opencl_path_tracer_amd import scenes
    W = H = 32
    fov = 2.0
    objs = [(quad(1000.0), scenes.CHROMIUM), (quad(-3000.0), scenes.LAMP)]
    sc = build(api, W, H, fov, objs)
    sc.set_option("variant", variant)
    sc.iterations = 2
    for n in (5, 11, 8):
        sc.render(n)
    got = sc.read_colors()[:, :3]
    one = build(api, W, H, fov, objs)
    one.set_option("variant", variant)
    one.iterations = 2
    one.render(24)
    assert np.array_equal(got, one.read_colors()[:, :3]) and np.array_equal(sc.read_rnds(), one.read_rnds())
    n_, k_ = np.array([3.10, 3.05, 2.05], np.float32), np.array([3.3, 3.3, 2.9], np.float32)
    F0 = ((k_ * k_ + (n_ - 1) * (n_ - 1)) / (k_ * k_ + (n_ + 1) * (n_ + 1))).astype(np.float64)
    state = seeds(W * H)
    ids = np.arange(W * H)
    vals = []
    for s in range(24):
        state, r1 = draw32(state)
        state, r2 = draw32(state)
        vals.append(camera_dir(W, H, fov, ids, r1, r2)[:, 2])
        state, _ = draw32(state)
        state, _ = draw32(state)
    run = np.zeros(W * H)
    for s, v in enumerate(vals):                                          # the recurrence itself, in float64
        run = (run * s + v) / (s + 1)
    assert np.allclose(run, np.mean(vals, axis=0), rtol=1e-12)
    exp = 2.0 * LAMP_E[None, :] * F0[None, :] * run[:, None]
    assert np.allclose(got.astype(np.float64), exp, rtol=2e-5, atol=0)
    # and a prefix: after the first two launches the frame is the mean of the first 16 values
    two = build(api, W, H, fov, objs)
    two.set_option("variant", variant)
    two.iterations = 2
    two.render(5)
    two.render(11)
    exp16 = 2.0 * LAMP_E[None, :] * F0[None, :] * np.mean(vals[:16], axis=0)[:, None]
    assert np.allclose(two.read_colors()[:, :3].astype(np.float64), exp16, rtol=2e-5, atol=0)
