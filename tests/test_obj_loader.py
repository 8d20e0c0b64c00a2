"""-m "not gpu": pt_add_obj (own OBJ/MTL reader) against fixtures produced by the
REFERENCE's own vendored parser.  tests/golden/obj/*.tinyobj.json is the dump of
tinyobj::LoadObj (/root/reference/tiny_obj_loader.h compiled into oracle/_ref/tinyobj_dump;
generator: tests/golden/make_obj_golden.py).  The expected triangles are built from that dump by
applying Scene::add_Obj's own steps (main.cpp:562-616) with the oracle's constructors."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "golden", "obj", "scene1.obj")


def expected_from_dump(oracle, dump, pos, scale, pitch, yaw, mat_offset=0):
    mats = []
    for m in dump["materials"]:                                   # main.cpp:564-572
        def f3(s):
            parts = s.split(" ")
            return [np.float32(float(parts[i])) if parts[i] else np.float32(0) for i in range(3)]
        kn, kk, tp = m["unknown"]["Kn"], m["unknown"]["Kk"], int(m["unknown"]["Tp"])
        mats.append(oracle.make_material(m["diffuse"], m["specular"], m["emission"], f3(kn), f3(kk), m["shininess"], tp)[0])
    V = np.array(dump["vertices"], dtype=np.float32).reshape(-1, 3)
    tris, obj_begin = [], []
    for sh in dump["shapes"]:                                     # main.cpp:587-616
        obj_begin.append(len(tris))
        off = 0
        for f, fv in enumerate(sh["num_face_vertices"]):
            vs = [oracle.obj_vertex(V[sh["vertex_index"][off + k]], pos, scale, pitch, yaw) for k in range(3)]
            off += fv
            tris.append(oracle.make_triangle(vs[0], vs[1], vs[2], mat_offset + sh["material_ids"][f])[0])
    return np.array(tris), np.array(mats), obj_begin


@pytest.mark.parametrize("pos,scale,pitch,yaw", [((0, 0, 0), (1, 1, 1), 0.0, 0.0), ((50, 330, -150), (190, 190, 190), -90.0, 50.0)])
def test_add_obj_matches_reference_parser(api, oracle, pos, scale, pitch, yaw):
    dump = json.load(open(OBJ[:-4] + ".tinyobj.json"))
    assert dump["ret"] is True and len(dump["shapes"]) == 3 and len(dump["materials"]) == 4
    sc = api.Scene(16, 16, device=None)
    sc.add_Obj(OBJ, pos, scale, pitch, yaw)
    tris, mats, objs = sc.debug_scene()
    etris, emats, eobjs = expected_from_dump(oracle, dump, pos, scale, pitch, yaw)
    assert objs.tolist() == eobjs
    assert mats.tobytes() == emats.astype(api.MATERIAL).tobytes()
    assert tris.shape[0] == etris.shape[0] == 11
    assert tris.tobytes() == etris.astype(api.TRIANGLE).tobytes()
    # fan triangulation of the quad and the pentagon, negative indices, first "Kn" wins
    assert [int(t["mati"]) for t in tris] == [0, 0, 3, 3, 3, 1, 1, 2, 2, 2, 0]
    sc.upload_Triangles()
    sc.upload_Materials()


def test_add_obj_offsets_materials_after_existing_ones(api, oracle):
    dump = json.load(open(OBJ[:-4] + ".tinyobj.json"))
    sc = api.Scene(16, 16, device=None)
    sc.add_Material((0.1, 0.2, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)
    sc.add_Material((0.1, 0.2, 0.3), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), 50.0, 0)
    sc.add_Obj(OBJ, (1, 2, 3), (2, 2, 2), 10.0, 20.0)
    tris, mats, _ = sc.debug_scene()
    etris, _, _ = expected_from_dump(oracle, dump, (1, 2, 3), (2, 2, 2), 10.0, 20.0, mat_offset=2)   # main.cpp:562
    assert mats.shape[0] == 6 and tris.tobytes() == etris.astype(api.TRIANGLE).tobytes()


def test_add_obj_errors_are_codes_not_exit(api, tmp_path):
    sc = api.Scene(16, 16, device=None)
    with pytest.raises(api.PtError) as e:
        sc.add_Obj(str(tmp_path / "missing.obj"), (0, 0, 0), (1, 1, 1), 0, 0)     # reference: exit(1), main.cpp:560
    assert e.value.code == api.PT_EIO
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\n")                       # no Kn/Kk/Tp: .at() throws, main.cpp:568
    (tmp_path / "a.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    with pytest.raises(api.PtError) as e:
        sc.add_Obj(str(tmp_path / "a.obj"), (0, 0, 0), (1, 1, 1), 0, 0)
    assert e.value.code == api.PT_EIO and "Kn" in str(e.value)
    (tmp_path / "b.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")      # face without material
    with pytest.raises(api.PtError):
        sc.add_Obj(str(tmp_path / "b.obj"), (0, 0, 0), (1, 1, 1), 0, 0)
