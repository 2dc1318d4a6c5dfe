#!/usr/bin/env python3
"""The depth pre-processing stage alone (tsdf_set_depth_frame, 640x480, PCL-like defaults), for
`rocprofv3 --kernel-trace --stats -- python3 tools/preproc_workload.py [grid_filter]` (tools/collect_profiles.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import tracking_sdf_amd as ts  # noqa: E402
from tracking_sdf_amd import synth  # noqa: E402

grid_filter = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seq = synth.Sequence(n_frames=1, width=640, height=480, noise=True, holes=0.02, step=5)
s = ts.SDF(64)
t = ts.CameraTracking(sdf=s)
t.set_K(seq.K)
xyz, nrm, rgb = seq.frame(0)
d16 = np.where(np.isnan(xyz[..., 2]), 0, np.round(xyz[..., 2] * 5000.0)).astype(np.uint16)
for _ in range(40):
    s.set_depth_frame(d16, rgb, grid_filter=grid_filter)
s.get_preprocessed()
s.close()
