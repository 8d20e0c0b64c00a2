"""Synthetic scene DATA for the BASELINE configs.

No reference assets exist (its OBJ models live outside the repository, main.cpp:1002-1010),
so the scenes are generated here.  Coordinates and material parameters of the Cornell box
are the ones the reference's scene-authoring code lists (cited per item); everything else
is a deterministic generator.
"""
from dataclasses import dataclass, field

import numpy as np

# (kd, ks, emission, N, K, shininess, type) -- the ten built-in materials, main.cpp:753-762
LAMP, SUN, WHITE_DIFFUSE, RED_DIFFUSE, GREEN_DIFFUSE, PURPLE_SPECULAR, BLACK_SPECULAR, CHROMIUM, GOLD, GLASS = range(10)
BUILTIN_MATERIALS = [
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (60.0 * 2, 50.0 * 2, 40.0 * 2), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, 3),
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (60.0 * 5, 50.0 * 5, 40.0 * 5), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 0.0, 3),
    ((0.3, 0.3, 0.3), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 50.0, 0),
    ((0.3, 0.1, 0.1), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 50.0, 0),
    ((0.1, 0.3, 0.1), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 50.0, 0),
    ((0.3, 0.0, 0.0), (0.3, 0.3, 0.3), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 200.0, 0),
    ((0.05, 0.05, 0.05), (0.3, 0.3, 0.3), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), 200.0, 0),
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (3.10, 3.05, 2.05), (3.3, 3.3, 2.9), 0.0, 1),
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.17, 0.35, 1.50), (3.1, 2.7, 1.9), 0.0, 1),
    ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.50, 1.50, 1.50), (0.0, 0.0, 0.0), 0.0, 2),
]


@dataclass
class SceneSpec:
    materials: list
    objects: list = field(default_factory=list)      # [(verts (n,3,3) f32, mati (n,) u16)] -- one end_Obj each
    fov: float = 60.0                                # the reference's "canonical" view, main.cpp:33-35,40
    yaw: float = 0.0
    pitch: float = 0.0
    shift: tuple = (0.0, 0.0, 0.0)
    name: str = ""

    @property
    def ntris(self):
        return sum(v.shape[0] for v, _ in self.objects)


def _quad_tris(rows, mat):
    v = np.asarray(rows, dtype=np.float32).reshape(-1, 3, 3)
    return v, np.full(v.shape[0], mat, dtype=np.uint16)


def cornell_walls():
    """6 quads = 12 triangles, vertex order as listed in the reference."""
    rows, mats = [], []

    def tri(a, b, c, m):
        rows.append((a, b, c))
        mats.append(m)

    # lamp, main.cpp:765-766
    tri((300.0, 999.9, 700.0), (300.0, 999.9, 300.0), (700.0, 999.9, 700.0), LAMP)
    tri((700.0, 999.9, 700.0), (300.0, 999.9, 300.0), (700.0, 999.9, 300.0), LAMP)
    # far wall z=1000, main.cpp:794-795
    tri((-100.0, 0.0, 1000.0), (-100.0, 1000.0, 1000.0), (1100.0, 1000.0, 1000.0), WHITE_DIFFUSE)
    tri((1100.0, 1000.0, 1000.0), (1100.0, 0.0, 1000.0), (-100.0, 0.0, 1000.0), WHITE_DIFFUSE)
    # left wall x=-100, main.cpp:798-799
    tri((-100.0, 0.0, 1000.0), (-100.0, 0.0, -1000.0), (-100.0, 1000.0, 1000.0), RED_DIFFUSE)
    tri((-100.0, 1000.0, 1000.0), (-100.0, 0.0, -1000.0), (-100.0, 1000.0, -1000.0), RED_DIFFUSE)
    # right wall x=1100, main.cpp:802-803
    tri((1100.0, 1000.0, 1000.0), (1100.0, 0.0, -1000.0), (1100.0, 0.0, 1000.0), GREEN_DIFFUSE)
    tri((1100.0, 1000.0, -1000.0), (1100.0, 0.0, -1000.0), (1100.0, 1000.0, 1000.0), GREEN_DIFFUSE)
    # ceiling y=1000, main.cpp:806-807
    tri((-100.0, 1000.0, 1000.0), (-100.0, 1000.0, -1000.0), (1100.0, 1000.0, 1000.0), WHITE_DIFFUSE)
    tri((1100.0, 1000.0, 1000.0), (-100.0, 1000.0, -1000.0), (1100.0, 1000.0, -1000.0), WHITE_DIFFUSE)
    # floor y=0, main.cpp:814-815
    tri((-10000.0, 0.0, -10000.0), (-10000.0, 0.0, 10000.0), (10000.0, 0.0, 10000.0), WHITE_DIFFUSE)
    tri((10000.0, 0.0, 10000.0), (10000.0, 0.0, -10000.0), (-10000.0, 0.0, -10000.0), WHITE_DIFFUSE)
    return np.asarray(rows, dtype=np.float32), np.asarray(mats, dtype=np.uint16)


def uv_sphere(center, radius, segments=32, rings=16):
    """Tessellated sphere (the reference has no analytic sphere: prog.cl:18-21).
    segments*2 cap triangles + (rings-2)*segments*2 band triangles."""
    cx, cy, cz = center
    tris = []

    def p(i, j):
        th = np.pi * i / rings
        ph = 2.0 * np.pi * (j % segments) / segments
        return (cx + radius * np.sin(th) * np.cos(ph), cy + radius * np.cos(th), cz + radius * np.sin(th) * np.sin(ph))

    for j in range(segments):
        tris.append((p(0, 0), p(1, j + 1), p(1, j)))
    for i in range(1, rings - 1):
        for j in range(segments):
            a, b, c, d = p(i, j), p(i, j + 1), p(i + 1, j), p(i + 1, j + 1)
            tris.append((a, b, d))
            tris.append((a, d, c))
    for j in range(segments):
        tris.append((p(rings, 0), p(rings - 1, j), p(rings - 1, j + 1)))
    return np.asarray(tris, dtype=np.float64).astype(np.float32)


def cornell_box(segments=32, rings=16):
    """Scene CB of SURVEY 8(d): 12 wall/lamp triangles + two tessellated spheres
    (radius 200; CHROMIUM at (250,200,300), GLASS at (750,200,-200)), 3 objects."""
    spec = SceneSpec(materials=list(BUILTIN_MATERIALS), name="cornell_box")
    spec.objects.append(cornell_walls())
    s1 = uv_sphere((250.0, 200.0, 300.0), 200.0, segments, rings)
    spec.objects.append((s1, np.full(s1.shape[0], CHROMIUM, dtype=np.uint16)))
    s2 = uv_sphere((750.0, 200.0, -200.0), 200.0, segments, rings)
    spec.objects.append((s2, np.full(s2.shape[0], GLASS, dtype=np.uint16)))
    return spec


def _lcg_floats(n, seed=1):
    """minstd (48271) stream mapped to [0,1): deterministic displacement noise."""
    out = np.empty(n, dtype=np.float64)
    x = seed
    for i in range(n):
        x = (x * 48271) % 2147483647
        out[i] = x / 2147483647.0
    return out


def displaced_grid_mesh(ntris_target, seed=1):
    """MESH-100k / MESH-1M of SURVEY 8(d): CB walls + one displaced height-field grid inside
    the box, material bands WHITE/CHROMIUM/GLASS by face index."""
    n = int(np.ceil(np.sqrt(ntris_target / 2.0)))
    xs = np.linspace(0.0, 1000.0, n + 1)
    zs = np.linspace(-600.0, 800.0, n + 1)
    rng = np.random.RandomState(seed)           # deterministic; numpy's MT19937 is stable across versions
    h = rng.rand(n + 1, n + 1)
    gx, gz = np.meshgrid(xs, zs, indexing="ij")
    gy = 80.0 + 120.0 * (np.sin(gx / 97.0) * np.cos(gz / 131.0) + 1.0) + 25.0 * h
    P = np.stack([gx, gy, gz], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    t1 = np.stack([a, c, d], axis=2).reshape(-1, 3, 3)
    t2 = np.stack([a, d, b], axis=2).reshape(-1, 3, 3)
    verts = np.empty((t1.shape[0] * 2, 3, 3), dtype=np.float32)
    verts[0::2] = t1
    verts[1::2] = t2
    nt = verts.shape[0]
    band = (np.arange(nt) * 3) // nt
    mati = np.choose(band, [WHITE_DIFFUSE, CHROMIUM, GLASS]).astype(np.uint16)
    spec = SceneSpec(materials=list(BUILTIN_MATERIALS), name="mesh_%d" % nt)
    spec.objects.append(cornell_walls())
    spec.objects.append((verts, mati))
    return spec


# ------------------------------------------------------------------------------------------------
# MESH-* as files: SURVEY 8(d) specifies the mesh configs "written as OBJ+MTL with Kd/Ks/Ke/Ns/Kn/Kk/Tp
# and loaded through the build's loader with add_Obj semantics" (main.cpp:552-617).
def _mtl_block(name, m):
    kd, ks, ke, N, K, shininess, mtype = m
    return ("newmtl %s\nKd %r %r %r\nKs %r %r %r\nKe %r %r %r\nNs %r\nKn %r %r %r\nKk %r %r %r\nTp %d\n\n" %
            ((name,) + tuple(float(x) for x in kd) + tuple(float(x) for x in ks) + tuple(float(x) for x in ke) + (float(shininess),) +
             tuple(float(x) for x in N) + tuple(float(x) for x in K) + (int(mtype),)))


def write_grid_mesh_obj(ntris_target, directory, pos=(0.0, 0.0, 0.0), scale=(1.0, 1.0, 1.0), pitch=0.0, yaw=0.0, name=None, seed=1):
    """The height field of displaced_grid_mesh() as <name>.obj + <name>.mtl with SHARED vertices, one shape,
    three `usemtl` bands (WHITE_DIFFUSE, CHROMIUM, GLASS).  The file holds the coordinates add_Obj's
    transform (negate x, rotate_x(pitch), rotate_y(yaw), * scale + pos: main.cpp:598-606) maps back onto the
    height field -- up to rounding: what the loader must reproduce bit for bit is ITS OWN arithmetic on the
    file's numbers, which the tests restate with the oracle.  Returns (obj_path, vertices (nv,3) f32 as
    written, faces (nt,3) int vertex indices, material index per face relative to the file's materials)."""
    import os
    spec = displaced_grid_mesh(ntris_target, seed)
    verts = spec.objects[1][0].astype(np.float64)                 # (nt, 3, 3) world coordinates
    nt = verts.shape[0]
    flat = verts.reshape(-1, 3)
    uniq, inv = np.unique(flat, axis=0, return_inverse=True)
    faces = inv.reshape(nt, 3)
    # inverse of the loader's transform, in double
    w = (uniq - np.asarray(pos, np.float64)) / np.asarray(scale, np.float64)
    b = np.float64(np.float32(yaw) / np.float32(180.0) * np.float32(3.141593))
    g = np.float64(np.float32(pitch) / np.float32(180.0) * np.float32(3.141593))
    x0 = w[:, 0] * np.cos(b) - w[:, 2] * np.sin(b)                # rotate_y(-yaw)
    z0 = w[:, 0] * np.sin(b) + w[:, 2] * np.cos(b)
    y1 = w[:, 1] * np.cos(g) + z0 * np.sin(g)                     # rotate_x(-pitch)
    z1 = -w[:, 1] * np.sin(g) + z0 * np.cos(g)
    local = np.stack([-x0, y1, z1], 1).astype(np.float32)
    name = name or ("mesh_%d" % nt)
    band_mats = [WHITE_DIFFUSE, CHROMIUM, GLASS]
    band = (np.arange(nt) * 3) // nt
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, name + ".mtl"), "w") as f:
        for k, mi in enumerate(band_mats):
            f.write(_mtl_block("band%d" % k, BUILTIN_MATERIALS[mi]))
    path = os.path.join(directory, name + ".obj")
    with open(path, "w") as f:
        f.write("# displaced grid, %d triangles, %d vertices\nmtllib %s.mtl\no %s\n" % (nt, local.shape[0], name, name))
        f.write("".join("v %r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])) for v in local))
        for k in range(3):
            f.write("usemtl band%d\n" % k)
            sel = faces[band == k] + 1
            f.write("".join("f %d %d %d\n" % (a, b_, c) for a, b_, c in sel))
    return path, local, faces, band.astype(np.uint16)
