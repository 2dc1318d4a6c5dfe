"""Worker of tests/test_distributed_gloo.py: one rank of the x-slab sharded tracker on CPU.

The per-slab accumulation is done by the CPU oracle (there is no GPU here); everything else is the
product's own multi-rank logic: tsdf_slab_range for the partition, the 30-double reduction row of
tsdf_device.h, a torch.distributed (gloo) sum all-reduce in the role of the RCCL call, and
tsdf_host_gn_step for the solve / exponential map / stop rule / pose update every rank replays.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle as orc            # noqa: E402
import tracking_sdf_amd as ts   # noqa: E402
from tracking_sdf_amd import synth  # noqa: E402
from util import make_oracle    # noqa: E402


def pack_row(A, b, st):
    row = np.zeros(ts.RED_ALLREDUCE)
    e = 0
    for a in range(6):
        for c in range(a, 6):
            row[e] = A[a, c]
            e += 1
    row[21:27] = b
    row[27] = st["n_terms"]
    row[29] = st["n_ok"]
    return row


def unpack_row(row):
    A = np.zeros((6, 6))
    e = 0
    for a in range(6):
        for c in range(a, 6):
            A[a, c] = A[c, a] = row[e]
            e += 1
    return A, row[21:27].copy()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, w, h = 32, 96, 72
    seq = synth.Sequence(n_frames=4, width=w, height=h, noise=True, holes=0.03, step=3)
    frames = [seq.frame(k) for k in range(4)]
    oo, ot = make_oracle(m, seq.K)
    for k in range(3):
        ot.set_camera_transformation(seq.R[k], seq.t[k])
        oo.update(ot, orc.Cloud(*frames[k]))
    cloud = orc.Cloud(frames[3][0])
    x0, x1 = ts.slab_range(m, world, rank)
    rot, trans = seq.R[2].copy(), seq.t[2].copy()
    iters, stop, terms = 0, False, 0
    while iters < 20 and not stop:
        ot.set_camera_transformation(rot, trans)
        A, b, st = ot.accumulate(oo, cloud, threads=1, stale_carry=True, own_x0=x0, own_x1=x1)
        row = torch.from_numpy(pack_row(A, b, st))
        dist.all_reduce(row)                      # the RCCL all-reduce of the GPU build
        A, b = unpack_row(row.numpy())
        terms = int(row[27])
        rot, trans, tw, stop = ts.host_gn_step(rot, trans, A, b, 0.001)
        iters += 1
    out = {"rank": rank, "slab": [x0, x1], "iters": iters, "stop": stop, "terms": terms,
           "rot": rot.tolist(), "trans": trans.tolist()}
    # every rank must hold the identical pose (same reduced row, same deterministic host code)
    poses = [None] * world
    dist.all_gather_object(poses, (rot.tolist(), trans.tolist()))
    out["identical_across_ranks"] = all(p == poses[0] for p in poses)
    if rank == 0:
        # single-process reference: the oracle's own estimate_new_position
        ot.set_camera_transformation(seq.R[2], seq.t[2])
        st = ot.estimate_new_position(oo, cloud, threads=1, stale_carry=True)
        out["ref_iters"] = st["iterations"]
        out["ref_rot"] = ot.rot.tolist()
        out["ref_trans"] = ot.trans.tolist()
        out["ref_terms"] = st["n_terms_last"]
        print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
