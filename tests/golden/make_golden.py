"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/pt_oracle.c).

These fixtures are the ORACLE's own output, frozen: they pin the oracle (and through it the HIP
path) against accidental change.  They are NOT outputs of the reference: it ships no fixtures and
cannot be built or run in this image -- parity with a reference execution
is therefore "unpinned".
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O                       # noqa: E402
from opencl_path_tracer_amd import scenes               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    spec = scenes.cornell_box()
    sc = O.load_scene(spec)
    cam = O.make_camera(60.0, 0.0, 0.0, (0, 0, 0), 64, 64)
    fr = O.OracleFrame(64, 64)
    segs = fr.render(sc, cam, 4, 0, 4, nthreads=8)
    np.savez_compressed(os.path.join(HERE, "cb_64x64_b4_s4.npz"), colors=fr.colors()[:, :3].copy(),
                        rnds=fr.rnds().copy(), segments=np.int64(segs))
    # BASELINE config 1: Cornell box, 256x256, 4 bounces, 16 spp -- keep a checksum-sized summary
    cam = O.make_camera(60.0, 0.0, 0.0, (0, 0, 0), 256, 256)
    fr = O.OracleFrame(256, 256)
    segs = fr.render(sc, cam, 4, 0, 16, nthreads=8)
    cols = fr.colors()[:, :3].copy()
    np.savez_compressed(os.path.join(HERE, "cb_256x256_b4_s16_summary.npz"),
                        crop=cols.reshape(256, 256, 3)[96:160, 96:160].copy(),
                        rnds_crc=np.uint32(np.bitwise_xor.reduce(fr.rnds().view(np.uint32))),
                        colors_sum=cols.astype(np.float64).sum(0), segments=np.int64(segs))


if __name__ == "__main__":
    main()
