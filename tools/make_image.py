"""Render a scene and write it through the C ABI's image writers (pt_write_ppm / pt_write_pfm); also a PNG
copy of the PPM (pure Python, zlib) because that is what repository browsers display.
  python tools/make_image.py out_prefix scene=cornell W=256 H=256 bounces=4 spp=16 [which=0]"""
import struct
import sys
import zlib

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from opencl_path_tracer_amd import api, scenes  # noqa: E402


def ppm_to_png(ppm, png):
    raw = open(ppm, "rb").read().split(b"\n", 3)
    W, H = [int(x) for x in raw[1].split()]
    rows = np.frombuffer(raw[3], dtype=np.uint8).reshape(H, W * 3)
    data = b"".join(b"\x00" + rows[y].tobytes() for y in range(H))

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    open(png, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) +
                          chunk(b"IDAT", zlib.compress(data, 9)) + chunk(b"IEND", b""))


if __name__ == "__main__":
    prefix = sys.argv[1]
    opts = dict(a.split("=") for a in sys.argv[2:])
    W, H = int(opts.get("W", 256)), int(opts.get("H", 256))
    bounces, spp, which = int(opts.get("bounces", 4)), int(opts.get("spp", 16)), int(opts.get("which", 0))
    scene = opts.get("scene", "cornell")
    spec = {"cornell": scenes.cornell_box, "mesh100k": lambda: scenes.displaced_grid_mesh(100000),
            "mesh1m": lambda: scenes.displaced_grid_mesh(1000000)}[scene]()
    sc = api.Scene(W, H).load(spec)
    sc.iterations = bounces
    done = 0
    while done < spp:
        n = min(256, spp - done)
        sc.render(n)
        done += n
    sc.write_ppm(prefix + ".ppm", which)
    sc.write_pfm(prefix + ".pfm")
    ppm_to_png(prefix + ".ppm", prefix + ".png")
    c = sc.read_colors()[:, :3]
    print("%s: %s %dx%d b%d spp%d mean radiance %s, black pixels %d" % (prefix, scene, W, H, bounces, spp, c.mean(0), int((c.sum(1) == 0).sum())))
