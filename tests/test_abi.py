"""-m "not gpu": the C-ABI library loads and exports every symbol include/pt_api.h declares;
calls that need a device fail loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported(api):
    lib = C.CDLL(os.path.join(ROOT, "opencl_path_tracer_amd", "libptamd.so"))
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libptamd.so does not export %s" % n
    assert sorted(api.EXPORTS) == names, "api.EXPORTS and pt_api.h disagree"


def test_library_carries_gfx950_code():
    blob = open(os.path.join(ROOT, "opencl_path_tracer_amd", "libptamd.so"), "rb").read()
    assert b"gfx950" in blob and b"k_render" in blob


def test_pod_sizes(api):
    assert api.MATERIAL.itemsize == 80 and api.TRIANGLE.itemsize == 80
    assert api.CAMERA.itemsize == 80 and api.RAY.itemsize == 32
    assert api.TRIANGLE.fields["mati"][1] == 64 and api.MATERIAL.fields["type"][1] == 72
    assert api.CAMERA.fields["XM"][1] == 64


def test_no_cpu_fallback(api, cb_spec):
    """A host-only context can author and build, but every render/readback call must refuse."""
    sc = api.Scene(32, 32, device=None).load(cb_spec)
    sc.iterations = 4
    for call in (lambda: sc.render(1), sc.generate_rays, sc.trace_rays, sc.read_colors, sc.read_rnds,
                 sc.seed_default, sc.sync, lambda: sc.resolve_ldr(0)):
        with pytest.raises(api.PtError) as e:
            call()
        assert e.value.code == api.PT_ENODEVICE
    sc.close()


def test_missing_device_is_loud(api):
    from conftest import have_gpu
    if have_gpu():
        pytest.skip("a GPU is present")
    with pytest.raises(api.PtError) as e:
        api.Scene(16, 16, device=0)
    assert e.value.code in (api.PT_ENODEVICE, api.PT_EHIP)


def test_call_order_errors(api):
    sc = api.Scene(16, 16, device=None)
    with pytest.raises(api.PtError):
        sc.end_Obj()                                   # empty object (main.cpp:216 would read tris[0])
    m = sc.add_Material((0.3, 0.3, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)
    assert m == 0
    sc.add_Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), 0)
    with pytest.raises(api.PtError):
        sc.upload_Triangles()                          # object not closed
    sc.end_Obj()
    sc.upload_Triangles()
    sc.add_Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), 7)
    sc.end_Obj()
    sc.upload_Triangles()
    with pytest.raises(api.PtError):
        sc.upload_Materials()                          # material 7 was never added
    # > 6 triangles sharing one centroid: the reference's build never terminates (main.cpp:246-257)
    sc2 = api.Scene(16, 16, device=None)
    sc2.add_Material((0.3, 0.3, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)
    for _ in range(7):
        sc2.add_Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), 0)
    with pytest.raises(api.PtError) as e:
        sc2.end_Obj()
    assert e.value.code == api.PT_ESCENE
