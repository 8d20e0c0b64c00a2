"""Host-side mirror of the reference's `Scene` class (main.cpp:363-742) over the C ABI of
libptamd.so (include/pt_api.h).  Same method names, same call order, same argument meaning:

    scene = Scene(width, height)            # Scene::init_Scene
    m = scene.add_Material(kd, ks, emission, N, K, shininess, type)
    scene.add_Triangle(r1, r2, r3, m); ...; scene.end_Obj()
    scene.upload_Triangles(); scene.upload_Materials()
    scene.render()                          # generate_rays + trace_rays, current_sample++

The reference keeps its parameters in globals (iterations, global_fov/yaw/pitch/shift,
current_sample: main.cpp:27-39); here they are attributes of the Scene object.

There is no CPU path: if libptamd.so is missing this module raises at import, and a Scene on a
machine without a gfx950 device raises at construction (device=None asks for a host-only
context that can author scenes and build the BVH but cannot render).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_LIB: an A/B build of the same library (`make ab`), for kernel experiments only
_LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(_HERE, "libptamd.so")

PT_OK, PT_EINVAL, PT_ENODEVICE, PT_EHIP, PT_ESCENE, PT_EIO, PT_ECOMM = 0, -1, -2, -3, -4, -5, -6
PT_COMM_ID_BYTES = 128

MATERIAL = np.dtype([("kd", "<f4", 4), ("ks", "<f4", 4), ("emission", "<f4", 4), ("F0", "<f4", 4),
                     ("n", "<f4"), ("shininess", "<f4"), ("type", "<i4"), ("_pad", "<i4")])
RAY = np.dtype([("P", "<f4", 4), ("D", "<f4", 4)])
# Node4q (csrc/pt_internal.hpp): byte k of a q word is child k's plane on the node's grid, plane = origin + q * 2^(exp - 127)
WIDE_NODE = np.dtype([("origin", "<f4", 3), ("exp", "u1", 3), ("nchild", "u1"), ("q", "u1", (6, 4)), ("spare", "<u4", 2), ("ref", "<i4", 4)])
assert WIDE_NODE.itemsize == 64
TRIANGLE = np.dtype([("r1", "<f4", 4), ("r2", "<f4", 4), ("r3", "<f4", 4), ("N", "<f4", 4),
                     ("mati", "<u2"), ("_pad", "u1", 14)])
CAMERA = np.dtype([("eye", "<f4", 4), ("lookat", "<f4", 4), ("up", "<f4", 4), ("right", "<f4", 4),
                   ("XM", "<f4"), ("YM", "<f4"), ("_pad", "<f4", 2)])
assert MATERIAL.itemsize == 80 and RAY.itemsize == 32 and TRIANGLE.itemsize == 80 and CAMERA.itemsize == 80

# every symbol include/pt_api.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "pt_material_init", "pt_triangle_init", "pt_triangles_init", "pt_camera_init", "pt_camera_move", "pt_create", "pt_create_tiled", "pt_destroy",
    "pt_last_error", "pt_device_info", "pt_add_material", "pt_add_triangle", "pt_add_triangles", "pt_end_obj",
    "pt_add_obj", "pt_upload_triangles", "pt_upload_materials", "pt_seed_default", "pt_upload_seeds",
    "pt_generate_rays", "pt_trace_rays", "pt_render", "pt_set_current_sample", "pt_get_current_sample", "pt_sync",
    "pt_local_pixel_count", "pt_local_pixel_ids", "pt_read_colors", "pt_read_rnds", "pt_read_rays",
    "pt_resolve_ldr", "pt_bind_framebuffer", "pt_device_colors", "pt_device_rnds", "pt_set_stream",
    "pt_set_option", "pt_get_stat", "pt_debug_bvh_sizes", "pt_debug_bvh_copy", "pt_debug_wide_nodes", "pt_debug_encounter_rank", "pt_debug_tile_cost", "pt_debug_launch_plan",
    "pt_debug_scene_sizes", "pt_debug_scene_copy", "pt_debug_closest_hit",
    "pt_slab_pixel_count", "pt_frame_size", "pt_comm_available", "pt_comm_unique_id", "pt_comm_init", "pt_gather_frame", "pt_device_frame", "pt_read_frame",
    "pt_write_pfm", "pt_write_ppm", "pt_image_write_pfm", "pt_image_write_ppm", "pt_debug_gather_index", "pt_debug_deinterleave",
]


class PtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libptamd: %s (code %d)" % (msg, code))
        self.code = code


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError("libptamd.so is not built (run `make` or __graft_entry__.build()); "
                          "there is no fallback implementation")
    L = C.CDLL(_LIB_PATH)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    fp = C.POINTER(C.c_float)

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("pt_material_init", None, vp, fp, fp, fp, fp, fp, f32, i32)
    sig("pt_triangle_init", None, vp, fp, fp, fp, C.c_uint16)
    sig("pt_triangles_init", None, vp, vp, vp, i64)
    sig("pt_camera_init", None, vp, f32, f32, f32, fp, i32, i32)
    sig("pt_camera_move", None, fp, f32, f32, f32, f32, f32)
    sig("pt_create", C.c_int, C.c_int, i32, i32, C.POINTER(vp))
    sig("pt_create_tiled", C.c_int, C.c_int, i32, i32, i32, i32, i32, C.POINTER(vp))
    sig("pt_destroy", None, vp)
    sig("pt_last_error", C.c_char_p, vp)
    sig("pt_device_info", C.c_int, vp, C.c_char_p, i32)
    sig("pt_add_material", C.c_int, vp, vp)
    sig("pt_add_triangle", C.c_int, vp, vp)
    sig("pt_add_triangles", C.c_int, vp, vp, i64)
    sig("pt_end_obj", C.c_int, vp)
    sig("pt_add_obj", C.c_int, vp, C.c_char_p, fp, fp, f32, f32)
    sig("pt_upload_triangles", C.c_int, vp)
    sig("pt_upload_materials", C.c_int, vp)
    sig("pt_seed_default", C.c_int, vp)
    sig("pt_upload_seeds", C.c_int, vp, vp, i64)
    sig("pt_generate_rays", C.c_int, vp, vp)
    sig("pt_trace_rays", C.c_int, vp, vp, i32, i32)
    sig("pt_render", C.c_int, vp, vp, i32, i32)
    sig("pt_set_current_sample", C.c_int, vp, i32)
    sig("pt_get_current_sample", C.c_int, vp, C.POINTER(i32))
    sig("pt_sync", C.c_int, vp)
    sig("pt_local_pixel_count", C.c_int, vp, C.POINTER(i64))
    sig("pt_local_pixel_ids", C.c_int, vp, vp, i64)
    sig("pt_read_colors", C.c_int, vp, vp, i64)
    sig("pt_read_rnds", C.c_int, vp, vp, i64)
    sig("pt_read_rays", C.c_int, vp, vp, i64)
    sig("pt_resolve_ldr", C.c_int, vp, i32, vp, i64)
    sig("pt_bind_framebuffer", C.c_int, vp, vp, vp)
    sig("pt_device_colors", vp, vp)
    sig("pt_device_rnds", vp, vp)
    sig("pt_set_stream", C.c_int, vp, vp)
    sig("pt_set_option", C.c_int, vp, C.c_char_p, i64)
    sig("pt_get_stat", C.c_int, vp, C.c_char_p, C.POINTER(C.c_double))
    sig("pt_debug_bvh_sizes", C.c_int, vp, C.POINTER(i64), C.POINTER(i64))
    sig("pt_debug_bvh_copy", C.c_int, vp, vp, vp, vp, vp)
    sig("pt_debug_wide_nodes", C.c_int, vp, vp, i64, C.POINTER(i64))
    sig("pt_debug_encounter_rank", C.c_int, vp, vp, i64)
    sig("pt_debug_tile_cost", C.c_int, vp, vp, i64)
    sig("pt_debug_launch_plan", C.c_int, vp, i32, i32, vp)
    sig("pt_debug_scene_sizes", C.c_int, vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64))
    sig("pt_debug_scene_copy", C.c_int, vp, vp, vp, vp)
    sig("pt_debug_closest_hit", C.c_int, vp, vp, i64, vp, vp)
    sig("pt_slab_pixel_count", C.c_int, vp, C.POINTER(i64))
    sig("pt_frame_size", C.c_int, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i64))
    sig("pt_comm_available", C.c_int)
    sig("pt_comm_unique_id", C.c_int, vp)
    sig("pt_comm_init", C.c_int, vp, vp)
    sig("pt_gather_frame", C.c_int, vp)
    sig("pt_device_frame", vp, vp)
    sig("pt_read_frame", C.c_int, vp, vp, i64)
    sig("pt_write_pfm", C.c_int, vp, C.c_char_p)
    sig("pt_write_ppm", C.c_int, vp, C.c_char_p, i32)
    sig("pt_image_write_pfm", C.c_int, C.c_char_p, vp, i32, i32)
    sig("pt_image_write_ppm", C.c_int, C.c_char_p, vp, i32, i32)
    sig("pt_debug_gather_index", C.c_int, i32, i32, i32, i32, i64, vp)
    sig("pt_debug_deinterleave", C.c_int, vp, vp, i64, vp)
    return L


LIB = _load()


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def Material(kd, ks, emission, N, K, shininess, mtype):
    """Material(kd,ks,emission,N,K,shininess,type), main.cpp:101-111 -> 80-byte record."""
    m = np.zeros(1, dtype=MATERIAL)
    LIB.pt_material_init(_ptr(m), _f3(kd), _f3(ks), _f3(emission), _f3(N), _f3(K), float(shininess), int(mtype))
    return m


def Triangle(r1, r2, r3, mati):
    """Triangle(r1,r2,r3,mati), main.cpp:144-166 -> 80-byte record."""
    t = np.zeros(1, dtype=TRIANGLE)
    LIB.pt_triangle_init(_ptr(t), _f3(r1), _f3(r2), _f3(r3), int(mati))
    return t


def Camera(fov, yaw, pitch, shift, width, height):
    """Camera(), main.cpp:311-347, with the globals it reads passed in."""
    c = np.zeros(1, dtype=CAMERA)
    LIB.pt_camera_init(_ptr(c), float(fov), float(yaw), float(pitch), _f3(shift), int(width), int(height))
    return c


def camera_move(shift, yaw, pitch, forward, rightward, upward):
    """main.cpp:334-336: the movement the reference's Camera() adds into global_shift; returns the new shift (3 floats)."""
    v = (C.c_float * 3)(*[float(x) for x in shift])
    LIB.pt_camera_move(v, float(yaw), float(pitch), float(forward), float(rightward), float(upward))
    return (v[0], v[1], v[2])


def triangles_from_vertices(verts, mati):
    """(n,3,3) float32 vertices + (n,) material indices -> (n,) TRIANGLE records."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 9)
    mati = np.ascontiguousarray(mati, dtype=np.uint16)
    out = np.zeros(verts.shape[0], dtype=TRIANGLE)
    LIB.pt_triangles_init(_ptr(out), _ptr(verts), _ptr(mati), verts.shape[0])
    return out


def comm_available():
    """(True, "") if librccl is bound in this process, else (False, why).  No collective is touched."""
    rc = LIB.pt_comm_available()
    return (True, "") if rc == PT_OK else (False, (LIB.pt_last_error(None) or b"").decode())


def comm_unique_id():
    """ncclGetUniqueId through the C ABI: 128 bytes for rank 0 to hand to every rank."""
    buf = (C.c_ubyte * PT_COMM_ID_BYTES)()
    rc = LIB.pt_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != PT_OK:
        raise PtError(rc, (LIB.pt_last_error(None) or b"").decode())
    return bytes(buf)


def gather_index(width, height, world, rows_per_block, slab_stride):
    out = np.empty(width * height, dtype=np.int64)
    rc = LIB.pt_debug_gather_index(width, height, world, rows_per_block, slab_stride, _ptr(out))
    if rc != PT_OK:
        raise PtError(rc, "pt_debug_gather_index")
    return out


def write_pfm(path, rgba, width, height):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32).reshape(width * height, 4)
    rc = LIB.pt_image_write_pfm(os.fsencode(path), _ptr(rgba), width, height)
    if rc != PT_OK:
        raise PtError(rc, "cannot write %s" % path)


def write_ppm(path, rgba, width, height):
    rgba = np.ascontiguousarray(rgba, dtype=np.float32).reshape(width * height, 4)
    rc = LIB.pt_image_write_ppm(os.fsencode(path), _ptr(rgba), width, height)
    if rc != PT_OK:
        raise PtError(rc, "cannot write %s" % path)


class Scene:
    """The reference's Scene (main.cpp:363-742) on one MI355X.

    device=None -> host-only context (authoring + BVH build only).  rank/world/rows_per_block
    select this context's interleaved row blocks of the global frame (multi-GPU tiling)."""

    def __init__(self, width, height, device=0, rank=0, world=1, rows_per_block=8):
        self.width, self.height = int(width), int(height)
        self.rank, self.world, self.rows_per_block = rank, world, rows_per_block
        # the reference's mutable globals (main.cpp:27-39), shipped defaults replaced by the
        # "canonical" view it keeps in comments (main.cpp:33-35, 40)
        self.iterations = 1
        self.fov, self.yaw, self.pitch, self.shift = 60.0, 0.0, 0.0, (0.0, 0.0, 0.0)
        self._h = C.c_void_p()
        dev = -1 if device is None else int(device)
        rc = LIB.pt_create_tiled(dev, self.width, self.height, rank, world, rows_per_block, C.byref(self._h))
        if rc != PT_OK:
            raise PtError(rc, (LIB.pt_last_error(None) or b"").decode())
        self.camera = Camera(self.fov, self.yaw, self.pitch, self.shift, self.width, self.height)

    # -- plumbing
    def _ck(self, rc):
        if rc < 0:
            raise PtError(rc, (LIB.pt_last_error(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_h", None):
            LIB.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def list_info(self):                                   # Scene::list_info, main.cpp:389-455
        buf = C.create_string_buffer(256)
        self._ck(LIB.pt_device_info(self._h, buf, 256))
        return buf.value.decode()

    # -- authoring (main.cpp:529-617)
    def add_Material(self, *args):
        m = args[0] if len(args) == 1 else Material(*args)
        return self._ck(LIB.pt_add_material(self._h, _ptr(m)))

    def add_Triangle(self, *args):
        t = args[0] if len(args) == 1 else Triangle(*args)
        self._ck(LIB.pt_add_triangle(self._h, _ptr(t)))

    def add_Triangles(self, tris):
        tris = np.ascontiguousarray(tris, dtype=TRIANGLE)
        self._ck(LIB.pt_add_triangles(self._h, _ptr(tris), tris.shape[0]))

    def end_Obj(self):
        self._ck(LIB.pt_end_obj(self._h))

    def add_Obj(self, file, pos, scale, pitch, yaw):
        self._ck(LIB.pt_add_obj(self._h, os.fsencode(file), _f3(pos), _f3(scale), float(pitch), float(yaw)))

    def upload_Triangles(self):
        self._ck(LIB.pt_upload_triangles(self._h))

    def upload_Materials(self):
        self._ck(LIB.pt_upload_materials(self._h))

    def load(self, spec):
        """Author a scenes.SceneSpec: materials, then one object per entry, then upload."""
        for m in spec.materials:
            self.add_Material(*m)
        for verts, mati in spec.objects:
            self.add_Triangles(triangles_from_vertices(verts, mati))
            self.end_Obj()
        self.upload_Triangles()
        self.upload_Materials()
        self.set_view(spec.fov, spec.yaw, spec.pitch, spec.shift)
        return self

    # -- camera / globals
    def set_view(self, fov, yaw, pitch, shift):
        self.fov, self.yaw, self.pitch, self.shift = fov, yaw, pitch, tuple(shift)
        self.camera = Camera(fov, yaw, pitch, shift, self.width, self.height)

    @property
    def current_sample(self):
        v = C.c_int32()
        self._ck(LIB.pt_get_current_sample(self._h, C.byref(v)))
        return v.value

    @current_sample.setter
    def current_sample(self, v):
        self._ck(LIB.pt_set_current_sample(self._h, int(v)))

    # -- the hot path (main.cpp:635-687)
    def generate_rays(self):
        self.camera = Camera(self.fov, self.yaw, self.pitch, self.shift, self.width, self.height)   # main.cpp:636
        self._ck(LIB.pt_generate_rays(self._h, _ptr(self.camera)))

    def trace_rays(self):
        self._ck(LIB.pt_trace_rays(self._h, _ptr(self.camera), self.iterations, self.current_sample))

    def render(self, nsamples=1, fused=True):
        """nsamples x Scene::render().  fused=False issues the reference's two launches per
        sample (generate_rays, trace_rays); fused=True is one persistent launch."""
        if fused:
            self._ck(LIB.pt_render(self._h, _ptr(self.camera), self.iterations, int(nsamples)))
        else:
            for _ in range(int(nsamples)):
                self.generate_rays()
                self.trace_rays()
                self.current_sample = self.current_sample + 1                                        # main.cpp:686

    def sync(self):
        self._ck(LIB.pt_sync(self._h))

    # -- seeds
    def seed_default(self):
        self._ck(LIB.pt_seed_default(self._h))

    def upload_seeds(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        self._ck(LIB.pt_upload_seeds(self._h, _ptr(seeds), seeds.size))

    # -- readback
    @property
    def local_pixels(self):
        v = C.c_int64()
        self._ck(LIB.pt_local_pixel_count(self._h, C.byref(v)))
        return v.value

    def local_pixel_ids(self):
        out = np.empty(self.local_pixels, dtype=np.int32)
        self._ck(LIB.pt_local_pixel_ids(self._h, _ptr(out), out.size))
        return out

    def read_colors(self):
        out = np.empty((self.local_pixels, 4), dtype=np.float32)
        self._ck(LIB.pt_read_colors(self._h, _ptr(out), out.shape[0]))
        return out

    def read_rnds(self):
        out = np.empty(self.local_pixels, dtype=np.int32)
        self._ck(LIB.pt_read_rnds(self._h, _ptr(out), out.size))
        return out

    def read_rays(self):
        out = np.empty(self.local_pixels, dtype=RAY)
        self._ck(LIB.pt_read_rays(self._h, _ptr(out), out.size))
        return out

    # -- frame assembly over RCCL (one process per GPU; the 128-byte id travels by the host's own means)
    @property
    def slab_pixels(self):
        v = C.c_int64()
        self._ck(LIB.pt_slab_pixel_count(self._h, C.byref(v)))
        return v.value

    def comm_init(self, id_bytes):
        buf = (C.c_ubyte * PT_COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        self._ck(LIB.pt_comm_init(self._h, C.cast(buf, C.c_void_p)))

    def gather_frame(self):
        self._ck(LIB.pt_gather_frame(self._h))

    def device_frame(self):
        return LIB.pt_device_frame(self._h)

    def read_frame(self):
        out = np.empty((self.width * self.height, 4), dtype=np.float32)
        self._ck(LIB.pt_read_frame(self._h, _ptr(out), out.shape[0]))
        return out

    # -- image files (what the reference shows through its GL blit, main.cpp:1019-1039)
    def write_pfm(self, path):
        self._ck(LIB.pt_write_pfm(self._h, os.fsencode(path)))

    def write_ppm(self, path, which=0):
        self._ck(LIB.pt_write_ppm(self._h, os.fsencode(path), int(which)))

    def debug_deinterleave(self, gathered):
        gathered = np.ascontiguousarray(gathered, dtype=np.float32).reshape(-1, 4)
        out = np.empty((self.width * self.height, 4), dtype=np.float32)
        self._ck(LIB.pt_debug_deinterleave(self._h, _ptr(gathered), gathered.shape[0], _ptr(out)))
        return out

    def resolve_ldr(self, which=0):
        out = np.empty((self.local_pixels, 4), dtype=np.float32)
        self._ck(LIB.pt_resolve_ldr(self._h, int(which), _ptr(out), out.shape[0]))
        return out

    # -- options / stats / device plumbing
    def set_option(self, key, value):
        self._ck(LIB.pt_set_option(self._h, key.encode(), int(value)))

    def stat(self, key):
        v = C.c_double()
        self._ck(LIB.pt_get_stat(self._h, key.encode(), C.byref(v)))
        return v.value

    def bind_framebuffer(self, colors_ptr, rnds_ptr):
        self._ck(LIB.pt_bind_framebuffer(self._h, C.c_void_p(colors_ptr), C.c_void_p(rnds_ptr)))

    def set_stream(self, stream_ptr):
        self._ck(LIB.pt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def device_colors(self):
        return LIB.pt_device_colors(self._h)

    def device_rnds(self):
        return LIB.pt_device_rnds(self._h)

    # -- introspection (host data)
    def debug_bvh(self):
        nn, nt = C.c_int64(), C.c_int64()
        self._ck(LIB.pt_debug_bvh_sizes(self._h, C.byref(nn), C.byref(nt)))
        nodes = np.zeros((nn.value, 16), dtype=np.float32)
        tris = np.zeros((nt.value, 12), dtype=np.float32)
        meta = np.zeros((nt.value, 2), dtype=np.int32)
        orig = np.zeros(nt.value, dtype=np.int32)
        self._ck(LIB.pt_debug_bvh_copy(self._h, _ptr(nodes), _ptr(tris), _ptr(meta), _ptr(orig)))
        return nodes, tris, meta, orig

    def debug_wide_nodes(self):
        """The 4-wide quantised nodes (pt_wide.cpp) as a structured array; empty when they were not built."""
        n = C.c_int64()
        self._ck(LIB.pt_debug_wide_nodes(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=WIDE_NODE)
        if n.value:
            self._ck(LIB.pt_debug_wide_nodes(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def debug_scene(self):
        nt, nm, no = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(LIB.pt_debug_scene_sizes(self._h, C.byref(nt), C.byref(nm), C.byref(no)))
        tris = np.zeros(nt.value, dtype=TRIANGLE)
        mats = np.zeros(nm.value, dtype=MATERIAL)
        objs = np.zeros(no.value, dtype=np.int32)
        self._ck(LIB.pt_debug_scene_copy(self._h, _ptr(tris), _ptr(mats), _ptr(objs)))
        return tris, mats, objs

    def debug_closest_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY)
        t = np.empty(rays.shape[0], dtype=np.float32)
        tri = np.empty(rays.shape[0], dtype=np.int32)
        self._ck(LIB.pt_debug_closest_hit(self._h, _ptr(rays), rays.shape[0], _ptr(t), _ptr(tri)))
        return t, tri

    def debug_encounter_rank(self, n):
        out = np.empty(n, dtype=np.int32)
        self._ck(LIB.pt_debug_encounter_rank(self._h, _ptr(out), n))
        return out

    def debug_launch_plan(self, nsamples, cu_count=0):
        """What render(nsamples) would launch on a device of cu_count compute units (host-only contexts too)."""
        out = np.zeros(8, dtype=np.int64)
        self._ck(LIB.pt_debug_launch_plan(self._h, int(nsamples), int(cu_count), _ptr(out)))
        keys = ("block", "waves_per_simd", "schedule", "chunk_spp", "resident_waves", "tiles", "node_mode", "lds_bytes")
        return dict(zip(keys, (int(v) for v in out)))

    def debug_tile_cost(self):
        """After set_option("count_work", 1) + render(n): per 8x8 tile of the local frame, the shader-clock cycles / 64 its wave spent on it."""
        local_rows = self.local_pixels // self.width
        n = ((self.width + 7) // 8) * ((local_rows + 7) // 8)
        out = np.empty(n, dtype=np.uint32)
        self._ck(LIB.pt_debug_tile_cost(self._h, _ptr(out), n))
        return out
