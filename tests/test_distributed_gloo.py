"""world_size-2 (and 3) run of the sharded tracker logic on CPU with the gloo backend."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tracker_matches_single_process(world):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT ")][0]
    r = json.loads(line[len("RESULT "):])
    assert r["identical_across_ranks"]
    assert r["iters"] == r["ref_iters"] and r["terms"] == r["ref_terms"] and r["terms"] > 100
    assert np.max(np.abs(np.array(r["rot"]) - np.array(r["ref_rot"]))) < 1e-10
    assert np.max(np.abs(np.array(r["trans"]) - np.array(r["ref_trans"]))) < 1e-10
