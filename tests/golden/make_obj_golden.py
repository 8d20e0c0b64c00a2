"""Regenerates tests/golden/obj/*.tinyobj.json by running the reference's own vendored parser
(/root/reference/tiny_obj_loader.h, compiled into oracle/_ref/tinyobj_dump by oracle/Makefile)
on the synthetic OBJ/MTL files next to them.  Only possible in the build container."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
exe = os.path.join(ROOT, "oracle", "_ref", "tinyobj_dump")
subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
for obj in sorted(glob.glob(os.path.join(HERE, "obj", "*.obj"))):
    out = subprocess.run([exe, obj], check=True, capture_output=True, text=True, cwd=os.path.join(HERE, "obj")).stdout
    open(obj[:-4] + ".tinyobj.json", "w").write(out)
    print("wrote", obj[:-4] + ".tinyobj.json")
