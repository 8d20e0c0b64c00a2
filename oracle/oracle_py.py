"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under opencl_path_tracer_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F3 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])
MATERIAL = np.dtype([("kd", "<f4", 4), ("ks", "<f4", 4), ("emission", "<f4", 4), ("F0", "<f4", 4),
                     ("n", "<f4"), ("shininess", "<f4"), ("type", "<i4"), ("_pad", "<i4")])
RAY = np.dtype([("P", "<f4", 4), ("D", "<f4", 4)])
TRIANGLE = np.dtype([("r1", "<f4", 4), ("r2", "<f4", 4), ("r3", "<f4", 4), ("N", "<f4", 4),
                     ("mati", "<u2"), ("_pad", "u1", 14)])
NODE = np.dtype([("trii", "<i4", 2), ("_pad", "<i4", 2), ("bl", "<f4", 4), ("tr", "<f4", 4)])
CAMERA = np.dtype([("eye", "<f4", 4), ("lookat", "<f4", 4), ("up", "<f4", 4), ("right", "<f4", 4),
                   ("XM", "<f4"), ("YM", "<f4"), ("_pad", "<f4", 2)])
HIT = np.dtype([("t", "<f4"), ("_p0", "<f4", 3), ("P", "<f4", 4), ("N", "<f4", 4), ("mati", "<u2"),
                ("_p1", "u1", 14), ("mat", MATERIAL)])
BBOX = np.dtype([("bl", "<f4", 4), ("tr", "<f4", 4)])
assert MATERIAL.itemsize == 80 and RAY.itemsize == 32 and TRIANGLE.itemsize == 80
assert NODE.itemsize == 48 and CAMERA.itemsize == 80 and HIT.itemsize == 144


def build(force=False):
    so = os.path.join(_HERE, "libpt_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("pt_oracle.c", "pt_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("PT_ORACLE_LIB") or os.path.join(_HERE, "libpt_oracle.so")     # (bench.py's cpu_baseline: the -march=native build)
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    fp = C.POINTER(C.c_float)
    ip = C.POINTER(C.c_int)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("orc_layout", i32, i32)
    sig("orc_seed_sequence", None, vp, i64)
    sig("orc_rand", f32, ip)
    sig("orc_spec_sincosf", None, f32, fp, fp)
    sig("orc_spec_powf", f32, f32, f32)
    sig("orc_spec_pow5", f32, f32)
    sig("orc_material_make", None, vp, fp, fp, fp, fp, fp, f32, i32)
    sig("orc_triangle_make", None, vp, fp, fp, fp, i32)
    sig("orc_camera_make", None, vp, f32, f32, f32, fp, i32, i32)
    sig("orc_camera_move", None, fp, f32, f32, f32, f32, f32)
    sig("orc_obj_vertex", None, fp, fp, fp, fp, f32, f32)
    sig("orc_scene_create", vp)
    sig("orc_scene_destroy", None, vp)
    sig("orc_add_material", i32, vp, vp)
    sig("orc_add_triangle", None, vp, vp)
    sig("orc_add_triangles", None, vp, vp, vp, i64)
    sig("orc_end_obj", i32, vp)
    sig("orc_scene_counts", i32, vp, ip, ip, ip, ip)
    sig("orc_scene_tris", vp, vp)
    sig("orc_scene_nodes", vp, vp)
    sig("orc_scene_shifts", vp, vp)
    sig("orc_scene_mats", vp, vp)
    sig("orc_scene_encounter_rank", None, vp, vp)
    sig("orc_camera_get_ray", None, vp, i32, vp, f32, f32)
    sig("orc_triangle_intersect", None, vp, vp, vp)
    sig("orc_bbox_intersection", i32, vp, vp, fp, fp)
    sig("orc_kd_intersect", None, vp, vp, vp, i32)
    sig("orc_new_ray_diffuse", None, vp, vp, vp, f32, f32)
    sig("orc_new_ray_specular", None, vp, vp, vp, vp)
    sig("orc_new_ray_refractive", None, vp, vp, vp, vp, f32, vp, ip, f32)
    sig("orc_fresnel", None, vp, vp, vp, vp)
    sig("orc_reinhard_tone_map", None, fp, fp)
    sig("orc_filmic_tone_map", None, fp, fp)
    sig("orc_frame_create", vp, i32, i32)
    sig("orc_frame_destroy", None, vp)
    sig("orc_frame_seed_default", None, vp)
    sig("orc_frame_rnds", vp, vp)
    sig("orc_frame_rays", vp, vp)
    sig("orc_frame_colors", vp, vp)
    sig("orc_frame_tex", vp, vp)
    sig("orc_gen_ray", None, vp, vp, i32)
    sig("orc_trace_ray", None, vp, vp, vp, i32, i32, i32, i32)
    sig("orc_render", i64, vp, vp, vp, i32, i32, i32, i32, i32)
    sig("orc_filt_im", None, vp, i32)
    _LIB = L
    return L


def _f3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


def _ptr(arr):
    return arr.ctypes.data_as(C.c_void_p)


def _view(addr, dtype, count):
    if not addr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (dtype.itemsize * count)).from_address(addr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def seed_sequence(n):
    out = np.empty(n, dtype=np.int32)
    lib().orc_seed_sequence(_ptr(out), n)
    return out


def make_material(kd, ks, em, N, K, shininess, mtype):
    m = np.zeros(1, dtype=MATERIAL)
    lib().orc_material_make(_ptr(m), _f3(kd), _f3(ks), _f3(em), _f3(N), _f3(K), float(shininess), int(mtype))
    return m


def make_triangle(r1, r2, r3, mati):
    t = np.zeros(1, dtype=TRIANGLE)
    lib().orc_triangle_make(_ptr(t), _f3(r1), _f3(r2), _f3(r3), int(mati))
    return t


def camera_move(shift, yaw, pitch, forward, rightward, upward):
    """Camera()'s side effect on global_shift, main.cpp:334-336; returns the new shift."""
    import ctypes as C
    v = (C.c_float * 3)(*[float(x) for x in shift])
    lib().orc_camera_move(v, float(yaw), float(pitch), float(forward), float(rightward), float(upward))
    return (v[0], v[1], v[2])


def make_camera(fov, yaw, pitch, shift, width, height):
    c = np.zeros(1, dtype=CAMERA)
    lib().orc_camera_make(_ptr(c), float(fov), float(yaw), float(pitch), _f3(shift), int(width), int(height))
    return c


def obj_vertex(v, pos, scale, pitch, yaw):
    out = (C.c_float * 3)()
    lib().orc_obj_vertex(out, _f3(v), _f3(pos), _f3(scale), float(pitch), float(yaw))
    return np.array(list(out), dtype=np.float32)


class OracleScene:
    """Mirror of the reference's Scene authoring calls (main.cpp:529-551)."""

    def __init__(self):
        self.h = lib().orc_scene_create()

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_Material(self, kd, ks, em, N, K, shininess, mtype):
        m = make_material(kd, ks, em, N, K, shininess, mtype)
        return lib().orc_add_material(self.h, _ptr(m))

    def add_Triangle(self, r1, r2, r3, mati):
        t = make_triangle(r1, r2, r3, mati)
        lib().orc_add_triangle(self.h, _ptr(t))

    def add_triangles(self, verts, mati):
        """verts: (n,3,3) float32, mati: (n,) ints -- one add_Triangle per row."""
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
        mati = np.ascontiguousarray(mati, dtype=np.uint16)
        lib().orc_add_triangles(self.h, _ptr(verts), _ptr(mati), verts.shape[0])

    def end_Obj(self):
        rc = lib().orc_end_obj(self.h)
        if rc != 0:
            raise RuntimeError("orc_end_obj failed: %d" % rc)

    def counts(self):
        a, b, c, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().orc_scene_counts(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(ntris=a.value, nnodes=b.value, nobj=c.value, nmats=d.value)

    def tris(self):
        return _view(lib().orc_scene_tris(self.h), TRIANGLE, self.counts()["ntris"]).copy()

    def nodes(self):
        n = self.counts()["nnodes"]
        return _view(lib().orc_scene_nodes(self.h), NODE, max(n, 0)).copy()

    def shifts(self):
        return _view(lib().orc_scene_shifts(self.h), np.dtype("<i4"), self.counts()["nobj"]).copy()

    def mats(self):
        return _view(lib().orc_scene_mats(self.h), MATERIAL, self.counts()["nmats"]).copy()

    def encounter_rank(self):
        out = np.empty(self.counts()["ntris"], dtype=np.int32)
        lib().orc_scene_encounter_rank(self.h, _ptr(out))
        return out

    def closest_hit(self, rays, mode=0):
        rays = np.ascontiguousarray(rays, dtype=RAY)
        out = np.zeros(rays.shape[0], dtype=HIT)
        L = lib()
        for i in range(rays.shape[0]):
            L.orc_kd_intersect(out[i:i + 1].ctypes.data_as(C.c_void_p), self.h,
                               rays[i:i + 1].ctypes.data_as(C.c_void_p), mode)
        return out


class OracleFrame:
    """rays / rnds / colors buffers of the reference (main.cpp:508-527) + the two kernels."""

    def __init__(self, width, height, seed_default=True):
        self.W, self.H = width, height
        self.h = lib().orc_frame_create(width, height)
        if seed_default:
            lib().orc_frame_seed_default(self.h)

    def close(self):
        if self.h:
            lib().orc_frame_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _n(self):
        return self.W * self.H

    def rnds(self):
        return _view(lib().orc_frame_rnds(self.h), np.dtype("<i4"), self._n())

    def rays(self):
        return _view(lib().orc_frame_rays(self.h), RAY, self._n())

    def colors(self):
        return _view(lib().orc_frame_colors(self.h), np.dtype("<f4"), self._n() * 4).reshape(self._n(), 4)

    def tex(self):
        return _view(lib().orc_frame_tex(self.h), np.dtype("<f4"), self._n() * 4).reshape(self._n(), 4)

    def generate_rays(self, cam, nthreads=1):
        lib().orc_gen_ray(self.h, _ptr(cam), nthreads)

    def trace_rays(self, scene, cam, iterations, current_sample, mode=0, nthreads=1):
        lib().orc_trace_ray(self.h, scene.h, _ptr(cam), iterations, current_sample, mode, nthreads)

    def render(self, scene, cam, iterations, first_sample, nsamples, mode=0, nthreads=1):
        return lib().orc_render(self.h, scene.h, _ptr(cam), iterations, first_sample, nsamples, mode, nthreads)

    def filt_im(self, nthreads=1):
        lib().orc_filt_im(self.h, nthreads)


def load_scene(spec):
    """Author an OracleScene from an opencl_path_tracer_amd.scenes.SceneSpec."""
    sc = OracleScene()
    for m in spec.materials:
        sc.add_Material(*m)
    for verts, mati in spec.objects:
        sc.add_triangles(verts, mati)
        sc.end_Obj()
    return sc
