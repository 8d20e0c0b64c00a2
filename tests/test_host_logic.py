"""-m "not gpu": host logic of the product (constructors, reference encounter order, own BVH,
tiling) checked against the oracle.  Uses host-only contexts; nothing renders."""
import numpy as np
import pytest

import bvh_check


def test_constructors_match_oracle(api, oracle, cb_spec):
    for m in cb_spec.materials:
        assert api.Material(*m).tobytes() == oracle.make_material(*m).tobytes()
    verts, mati = cb_spec.objects[1]
    for i in (0, 1, 17, 500, 959):
        a = api.Triangle(verts[i, 0], verts[i, 1], verts[i, 2], mati[i])
        b = oracle.make_triangle(verts[i, 0], verts[i, 1], verts[i, 2], mati[i])
        assert a.tobytes() == b.tobytes()
    for args in ((60, 0, 0, (0, 0, 0), 256, 256), (60, 0, 0, (0, 0, 0), 1920, 1080),
                 (75.0, -13.800002 - 50, 5.599997 + 10, (265.055481, 162.305969, 360.414001), 1536, 864),   # main.cpp:20-21,30-39
                 (33.3, 123.0, -77.0, (-5.5, 9.25, 1e3), 640, 480)):
        assert api.Camera(*args).tobytes() == oracle.make_camera(*args).tobytes()


def test_encounter_rank_matches_oracle(api, oracle, cb_spec, cb_oracle_scene):
    sc = api.Scene(16, 16, device=None).load(cb_spec)
    assert np.array_equal(sc.debug_encounter_rank(cb_spec.ntris), cb_oracle_scene.encounter_rank())


def test_encounter_rank_on_mesh(api, oracle):
    """The flattened, threaded restatement of NodeOnHost::build's leaf order (main.cpp:210-262) == the oracle's
    recursive one, for the serial path (small object) and with the top of the tree cut into parallel tasks."""
    from opencl_path_tracer_amd import scenes
    for ntris, threads in ((6000, 0), (60000, 1), (60000, 3), (60000, 0)):
        spec = scenes.displaced_grid_mesh(ntris)
        sc = api.Scene(16, 16, device=None)
        sc.set_option("build_threads", threads)
        sc.load(spec)
        osc = oracle.load_scene(spec)
        assert np.array_equal(sc.debug_encounter_rank(spec.ntris), osc.encounter_rank()), (ntris, threads)


def test_bvh_structure(api, cb_spec):
    sc = api.Scene(16, 16, device=None).load(cb_spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    assert tris.shape[0] == 1932 and sorted(orig.tolist()) == list(range(1932))
    assert sc.stat("flat_triangles") == 12          # the walls and the lamp: tested before the tree, in add order
    assert sc.stat("flat_boxes") == 6               # the two halves of a wall share a bounding box: one cull for both
    assert orig[:12].tolist() == list(range(12))
    depth = bvh_check.validate_structure(nodes, tris, 1932, 12)
    assert depth <= sc.stat("bvh_depth") <= 30
    # the traversal stack (LDS, [entry][lane]) is sized by the exact bound: sentinel + one far child per interior
    # level above the visited node + the slot a visit stores into above the top
    assert sc.stat("stack_entries") == ((depth + 2) + 1) & ~1
    # packets hold exactly the twelve floats prog.cl:94-112 reads, in add order via `orig`
    verts = np.concatenate([v for v, _ in cb_spec.objects]).reshape(-1, 9)
    assert np.array_equal(tris[:, :9], verts[orig])
    mati = np.concatenate([m for _, m in cb_spec.objects])
    assert np.array_equal(meta[:, 1], mati[orig])
    assert np.array_equal(meta[:, 0], sc.debug_encounter_rank(1932)[orig])
    assert sc.stat("node_mode") == 0 and sc.stat("treelet_nodes") == 0      # 941 nodes: the whole tree is staged in LDS


@pytest.mark.parametrize("ntris,want", [(6000, -1), (100000, -1), (100000, 300), (100000, 0)])
def test_treelet_reindexing(api, ntris, want):
    """Trees too large for LDS: the T nodes with the largest boxes are renumbered
    to [0, T) -- a connected top of the tree containing the root -- and the tree stays a valid BVH over
    all triangles.  T = what fits next to one 1,024-thread workgroup's stacks (or the requested count)."""
    from opencl_path_tracer_amd import scenes
    spec = scenes.displaced_grid_mesh(ntris)
    sc = api.Scene(16, 16, device=None)
    sc.set_option("treelet", want)
    sc.load(spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    T = int(sc.stat("treelet_nodes"))
    assert sorted(orig.tolist()) == list(range(spec.ntris))
    assert sc.stat("flat_triangles") == 12
    bvh_check.validate_structure(nodes, tris, spec.ntris, 12)
    if want == 0:             # the default
        assert T == 0 and sc.stat("node_mode") == 3          # no treelet: 4-wide nodes through L1/L2
        sc2 = api.Scene(16, 16, device=None)
        sc2.set_option("wide_nodes", 0)
        sc2.load(spec)
        assert sc2.stat("node_mode") == 1
        return
    assert sc.stat("node_mode") == 2
    entries = int(sc.stat("stack_entries"))
    assert bvh_check.validate_structure(nodes, tris, spec.ntris, 12) + 2 <= entries <= 36
    cap = (160 * 1024 - (32 * 100 + 4096 + 1024 + 256) - entries * 4 * 1024) // 64     # kLdsSlack: flat list + wf_intersect's arrays
    assert T == min(cap if want < 0 else min(cap, want), nodes.shape[0]) and T >= 2
    left, right = nodes[:, 12].view(np.int32), nodes[:, 13].view(np.int32)
    parent = np.full(nodes.shape[0], -1)
    for i in range(nodes.shape[0]):
        for c in (left[i], right[i]):
            if c >= 0:
                assert parent[c] == -1
                parent[c] = i
    assert parent[0] == -1 and (parent[1:] >= 0).all()
    assert (parent[1:T] < T).all(), "the treelet is not closed under 'parent'"
    # largest boxes first: no node outside the treelet whose parent is inside has a larger box than the
    # smallest treelet node
    lo = np.minimum(nodes[:, [0, 4, 8]], nodes[:, [2, 6, 10]])
    hi = np.maximum(nodes[:, [1, 5, 9]], nodes[:, [3, 7, 11]])
    d = (hi - lo).astype(np.float32)
    area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
    frontier = [i for i in range(T, nodes.shape[0]) if parent[i] < T]
    if frontier:
        assert area[frontier].max() <= area[:T].min()


@pytest.mark.parametrize("which", ["cornell", "mesh6k", "mesh100k"])
def test_wide_nodes_contain_the_bvh2(api, cb_spec, which):
    """4-wide quantised nodes (pt_wide.cpp): same leaves as the BVH2, every decoded child box contains the BVH2 boxes
    below it, and the reported stack bound is the one the structure implies."""
    from opencl_path_tracer_amd import scenes
    spec = cb_spec if which == "cornell" else scenes.displaced_grid_mesh(6000 if which == "mesh6k" else 100000)
    sc = api.Scene(16, 16, device=None)
    sc.set_option("wide_nodes", 2)
    sc.load(spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    wide = sc.debug_wide_nodes()
    assert wide.shape[0] == sc.stat("wide_nodes") > 0 and sc.stat("node_mode") == 3
    assert wide.shape[0] < 0.56 * nodes.shape[0]
    pending = bvh_check.validate_wide(nodes, wide, spec.ntris, int(sc.stat("flat_triangles")))
    assert pending == sc.stat("wide_pending")
    # default: only trees that do not fit LDS are collapsed
    sc2 = api.Scene(16, 16, device=None).load(spec)
    assert (sc2.stat("wide_nodes") > 0) == (which != "cornell")
    assert sc2.stat("node_mode") == (3 if which != "cornell" else 0)


def test_threaded_build_gives_the_same_tree(api):
    """The host SAH builder splits the top of the tree serially, builds the ranges below concurrently and splices them in
    preorder: node for node the tree of the single-threaded build (leaf references are positions, not append order)."""
    from opencl_path_tracer_amd import scenes
    spec = scenes.displaced_grid_mesh(100000)
    ref = None
    for threads in (1, 3, 8):
        sc = api.Scene(16, 16, device=None)
        sc.set_option("build_threads", threads)
        sc.load(spec)
        nodes, tris, meta, orig = sc.debug_bvh()
        if ref is None:
            ref = (nodes, tris, meta, orig, sc.stat("bvh_depth"))
            bvh_check.validate_structure(nodes, tris, spec.ntris, int(sc.stat("flat_triangles")))
            continue
        assert np.array_equal(nodes.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(orig, ref[3])
        assert np.array_equal(tris.view(np.uint32), ref[1].view(np.uint32)) and np.array_equal(meta, ref[2])
        assert sc.stat("bvh_depth") == ref[4]


def test_flat_list_selection(api, cb_spec):
    """The big-triangle list: the m biggest triangles, each >= 1/16 of the box around all the others.  Cornell
    box: the 10 wall / floor / ceiling triangles AND the 2 lamp triangles (the lamp is small, but not against
    the spheres' box); option flat_list caps it; a uniform mesh alone has none."""
    from opencl_path_tracer_amd import scenes
    import copy
    for cap, want in ((16, 12), (12, 12), (10, 10), (4, 4), (0, 0)):
        sc = api.Scene(16, 16, device=None)
        sc.set_option("flat_list", cap)
        sc.load(cb_spec)
        nodes, tris, meta, orig = sc.debug_bvh()
        assert sc.stat("flat_triangles") == want and sorted(orig.tolist()) == list(range(1932))
        bvh_check.validate_structure(nodes, tris, 1932, want)
        if want == 10:
            assert orig[:10].tolist() == list(range(2, 12))        # the lamp (triangles 0, 1) is the smallest of the twelve
    mesh_only = copy.deepcopy(scenes.displaced_grid_mesh(6000))
    mesh_only.objects = mesh_only.objects[1:]
    sc = api.Scene(16, 16, device=None).load(mesh_only)
    assert sc.stat("flat_triangles") == 0


def test_scene_size_cap(api):
    """Device offsets are 32-bit (packet index * 48, node index << 6): pt_add_triangles refuses more than
    2^26 triangles before it reads anything."""
    sc = api.Scene(8, 8, device=None)
    one = api.Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), 0)
    with pytest.raises(api.PtError) as e:
        api.LIB.pt_add_triangles.argtypes  # noqa: B018  (signature set by api._load)
        sc._ck(api.LIB.pt_add_triangles(sc._h, one.ctypes.data, (1 << 26) + 1))
    assert e.value.code == api.PT_EINVAL and "2^26" in str(e.value)
    sc._ck(api.LIB.pt_add_triangles(sc._h, one.ctypes.data, 1))           # the context is still usable


def test_option_validation(api):
    sc = api.Scene(8, 8, device=None)
    for key, bad in (("variant", 2), ("lds_scene", 1), ("lds_scene", 3), ("treelet", -2), ("treelet", 5000), ("chunk_spp", -2),
                     ("sah_visit_cost", -1), ("schedule", 3), ("schedule", -2), ("migrate_lanes", 0), ("migrate_lanes", 65), ("flat_list", -1), ("flat_list", 33), ("tile_order", 1), ("suspend_lanes", 64), ("suspend_lanes", -2), ("debug_repeat", -1), ("debug_repeat", 100000), ("bvh_policy", 6), ("bvh_device", 2), ("bvh_device", -2), ("sah_grain", 4), ("wide_on_device", 2),
                     ("block", 256), ("min_waves", 4), ("traversal", 1), ("pixel_map", 1), ("no_such_option", 0)):
        with pytest.raises(api.PtError) as e:
            sc.set_option(key, bad)
        assert e.value.code == api.PT_EINVAL, key


def test_bvh_never_culls_a_real_hit(api, oracle, cb_spec, cb_oracle_scene):
    """For random rays, the triangle the oracle's exhaustive search (mode 2) returns must lie in a
    leaf that the ray's box chain reaches in the product BVH (or in the big-triangle list in front of it)."""
    sc = api.Scene(16, 16, device=None).load(cb_spec)
    nodes, tris, meta, orig = sc.debug_bvh()
    n_flat = int(sc.stat("flat_triangles"))
    rng = np.random.RandomState(3)
    n = 300
    rays = np.zeros(n, dtype=oracle.RAY)
    P = np.stack([rng.uniform(-50, 1050, n), rng.uniform(10, 990, n), rng.uniform(-900, 900, n)], 1)
    D = rng.normal(size=(n, 3))
    D[:20, 0] = 0.0                       # axis-parallel components: division by zero in the slab test
    D[20:40, 1] = 0.0
    D /= np.linalg.norm(D, axis=1)[:, None]
    rays["P"][:, :3] = P
    rays["D"][:, :3] = D
    hits = cb_oracle_scene.closest_hit(rays, mode=2)
    otris = cb_oracle_scene.tris()
    checked = 0
    for i in range(n):
        if not hits[i]["t"] > 0:
            continue
        Pd, Dd = rays[i]["P"][:3].astype(np.float64), rays[i]["D"][:3].astype(np.float64)
        reach = bvh_check.leaves_reaching(nodes, Pd, Dd) + [(0, n_flat)]      # the flat list is tested by every ray
        best_t, found = float(hits[i]["t"]), False
        for first, count in reach:
            for k in range(first, first + count):
                t = bvh_check.tri_test(tris[k].astype(np.float64), Pd, Dd)
                if t > 0 and abs(t - best_t) <= 1e-4 * best_t:
                    found = True
        assert found, "ray %d: closest hit not reachable through the BVH" % i
        checked += 1
    assert checked > 200


def test_tiny_and_empty_scenes_build(api):
    sc = api.Scene(8, 8, device=None)
    sc.add_Material((0.3, 0.3, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)
    sc.upload_Triangles()                 # empty scene: a root with two empty children
    sc.upload_Materials()
    nodes, tris, meta, orig = sc.debug_bvh()
    assert nodes.shape[0] == 1 and tris.shape[0] == 0
    sc.add_Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), 0)
    sc.end_Obj()
    sc.upload_Triangles()
    nodes, tris, meta, orig = sc.debug_bvh()
    assert nodes.shape[0] == 1 and tris.shape[0] == 1
    assert sc.stat("flat_triangles") == 1 and bvh_check.validate_structure(nodes, tris, 1, 1) == 0


@pytest.mark.parametrize("H,world,rb", [(64, 1, 8), (64, 2, 8), (1080, 8, 8), (37, 4, 8), (100, 3, 16), (5, 8, 8)])
def test_tiling_partitions_the_frame(api, H, world, rb):
    """interleaved row blocks; every global pixel belongs to exactly one rank."""
    W = 24
    owner = np.full(W * H, -1)
    for r in range(world):
        sc = api.Scene(W, H, device=None, rank=r, world=world, rows_per_block=rb)
        ids = sc.local_pixel_ids()
        assert ids.size == sc.local_pixels
        assert (owner[ids] == -1).all()
        owner[ids] = r
        rows = np.unique(ids // W)
        assert ((rows // rb) % world == r).all()
    assert (owner >= 0).all()


def test_camera_movement_accumulates_like_the_reference(api, oracle):
    """main.cpp:334-336: every Camera() adds this frame's movement along the ROTATED unit axes into global_shift before the eye is
    placed; main.cpp:1189-1209 set global_forward / rightward / upward to +-speed * dt or 0.  Three frames of key presses (W, then
    W + D, then Q with a new yaw) replayed through pt_camera_move + pt_camera_init against the oracle's restatement: the shift and
    the 80 camera bytes agree bit for bit after every step, and a frame without movement leaves the shift alone."""
    fov, W, H = 75.0, 192, 108
    shift_p = shift_o = (265.055481, 162.305969, 360.414001)             # main.cpp:39
    steps = [(-63.800002, 15.599997, 1000.0 * 0.016, 0.0, 0.0),           # W for one 16-ms frame
             (-63.800002, 15.599997, 1000.0 * 0.021, 1000.0 * 0.021, 0.0),   # W + D
             (-20.0, 15.599997, 0.0, 0.0, 1000.0 * 0.033),                # Q after the mouse turned the view
             (-20.0, 15.599997, 0.0, 0.0, 0.0)]                           # keys released
    for yaw, pitch, fwd, rgt, upw in steps:
        before = shift_p
        shift_p = api.camera_move(shift_p, yaw, pitch, fwd, rgt, upw)
        shift_o = oracle.camera_move(shift_o, yaw, pitch, fwd, rgt, upw)
        assert np.array_equal(np.float32(shift_p).view(np.uint32), np.float32(shift_o).view(np.uint32))
        cam_p = api.Camera(fov, yaw, pitch, shift_p, W, H)
        cam_o = oracle.make_camera(fov, yaw, pitch, shift_o, W, H)
        assert cam_p.tobytes() == cam_o.tobytes()
        if fwd == rgt == upw == 0.0:
            assert shift_p == before
        else:
            assert shift_p != before
    # the movement is along the view: forward at yaw 0 / pitch 0 is +z only
    assert api.camera_move((0.0, 0.0, 0.0), 0.0, 0.0, 5.0, 0.0, 0.0) == (0.0, 0.0, 5.0)
    assert api.camera_move((1.0, 2.0, 3.0), 0.0, 0.0, 0.0, 2.0, -1.0) == (3.0, 1.0, 3.0)


def test_launch_plan_of_a_rank_of_eight(api, cb_spec):
    """The launch policy DESIGN.md section 6 states, pinned: on a 256-CU device a one-GPU 1080p Cornell frame takes the six-wave
    768-thread instance under the suspend schedule with passes of 32 samples (64 from 256 spp per launch); the rank of an 8-GPU job
    has 4,050 tiles for the 6,144 waves that instance would keep resident, so it gets the 512-thread instance, lockstep, whole
    tiles -- the measured-best shape for one tile per wave (profiles/r03/g_*, r04/i_*); so does a rank of 4 (8,160 tiles: fewer
    than two per wave of the wide instance), with passes of 8; a rank of 2 and a rank of a 4K frame over 8 (16,080 / 16,320 tiles)
    keep the wide instance in lockstep with passes of 8."""
    def plan(W, H, rank, world, nsamples=64):
        sc = api.Scene(W, H, device=None, rank=rank, world=world, rows_per_block=8).load(cb_spec)
        return sc.debug_launch_plan(nsamples, 256)
    one = plan(1920, 1080, 0, 1)
    assert (one["block"], one["waves_per_simd"], one["schedule"], one["chunk_spp"]) == (768, 6, 2, 32) and one["tiles"] == 32400 and one["resident_waves"] == 6144
    short = plan(1920, 1080, 0, 1, 16)         # short launches keep schedule 1 (no end worth tapering, and its lanes finish together)
    assert short["schedule"] == 1, short
    assert plan(1920, 1080, 0, 1, 256)["chunk_spp"] == 32
    for r in (0, 3, 7):
        p8 = plan(1920, 1080, r, 8)
        assert (p8["block"], p8["waves_per_simd"], p8["schedule"], p8["chunk_spp"]) == (512, 4, 0, 0), p8
        assert 3840 <= p8["tiles"] <= 4080 and p8["resident_waves"] == 4096 and p8["node_mode"] == 0
    p2, p4 = plan(1920, 1080, 1, 2), plan(1920, 1080, 2, 4)
    assert (p2["block"], p2["schedule"], p2["chunk_spp"]) == (768, 0, 8), p2
    assert (p4["block"], p4["waves_per_simd"], p4["schedule"], p4["chunk_spp"]) == (512, 4, 0, 8), p4
    k8 = plan(3840, 2160, 5, 8)             # 16,320 tiles: 2.7 per resident wave
    assert (k8["block"], k8["waves_per_simd"], k8["schedule"], k8["chunk_spp"]) == (768, 6, 0, 8), k8
    # nodes from global memory (as for a mesh): the restart schedule is the one whose lanes move on to the next work item, in short passes
    sc = api.Scene(1920, 1080, device=None)
    sc.set_option("lds_scene", 0)
    sc.load(cb_spec)
    m64, m16, m4 = sc.debug_launch_plan(64, 256), sc.debug_launch_plan(16, 256), sc.debug_launch_plan(4, 256)
    assert (m64["block"], m64["waves_per_simd"], m64["schedule"], m64["chunk_spp"]) == (256, 7, 2, 16) and m64["node_mode"] != 0, m64
    assert (m16["schedule"], m16["chunk_spp"]) == (2, 8) and (m4["schedule"], m4["chunk_spp"]) == (2, 0), (m16, m4)


def test_device_policies_fall_back_to_the_tree_they_name(api):
    """bvh_policy 5 is "policy 0's SAH tree, built on the device", and a build that cannot run there -- a host-only context,
    a scene with a non-finite triangle -- must come back with THAT tree from the host builder, node for node (round 3 fell back
    to policy 3's tree, leaves of <= 8 forced).  Policy 4 (device LBVH) has no host form: it falls back to policy 0's tree too."""
    from opencl_path_tracer_amd import scenes
    spec = scenes.displaced_grid_mesh(6000)
    nan_spec = scenes.displaced_grid_mesh(6000)
    v, m = nan_spec.objects[1]
    v = v.copy()
    v[1234, 1, 2] = np.nan
    nan_spec.objects[1] = (v, m)
    for s in (spec, nan_spec):
        ref = None
        for policy in (0, 5, 4):
            sc = api.Scene(64, 64, device=None)
            sc.set_option("bvh_policy", policy)
            sc.load(s)
            nodes, tris, meta, orig = sc.debug_bvh()
            got = (nodes.tobytes(), orig.tobytes(), int(sc.stat("bvh_depth")))
            if ref is None:
                ref = got
            assert got == ref, "policy %d on a host-only context is not policy 0's tree" % policy
    # ... and the forced-leaf policies still differ from it (the comparison above is not vacuous)
    sc = api.Scene(64, 64, device=None)
    sc.set_option("bvh_policy", 3)
    sc.load(spec)
    assert sc.debug_bvh()[0].tobytes() != ref[0]
