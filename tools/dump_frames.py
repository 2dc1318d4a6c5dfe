#!/usr/bin/env python3
"""Write n synthetic frames in the raw format tools/shim_demo.cpp reads."""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tracking_sdf_amd import synth  # noqa: E402


def dump(path, n=4, width=160, height=120, step=2, noise=True, holes=0.01):
    seq = synth.Sequence(n_frames=n, width=width, height=height, noise=noise, holes=holes, step=step)
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", n, width, height))
        f.write(np.ascontiguousarray(seq.K, dtype="<f8").tobytes())
        for k in range(n):
            xyz, nrm, rgb = seq.frame(k)
            f.write(struct.pack("<d", seq.stamps[k]))
            f.write(xyz.astype("<f4").tobytes())
            f.write(nrm.astype("<f4").tobytes())
            f.write(rgb.tobytes())
    return seq


if __name__ == "__main__":
    dump(sys.argv[1], *(int(a) for a in sys.argv[2:5]))
