"""Device-side exchange step (tsdf_comm_init_peer): several processes on the one GPU of the test box map each other's
exchange buffers through HIP IPC handles; the sums must come out in rank order on every rank, pass after pass, and a
rank whose peers never arrive must give up with TSDF_E_COMM instead of hanging the GPU."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import tracking_sdf_amd as ts
rank, nranks, name, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
s = ts.SDF(32)
s.comm_init_peer(nranks, rank, name)
out = {"rank": rank}
if mode == "sums":
    rows = []
    for k in range(64):                      # 64 exchanges: both slot parities, many times over
        v = np.array([(rank + 1) * 0.1 + k + e * 1e-3 for e in range(30)])
        rows.append(s.allreduce(v).tolist())
    out["rows"] = rows
    s.comm_finalize()
    out["after"] = s.allreduce(np.full(30, 7.0)).tolist()      # no exchange configured any more: identity
elif mode == "absent":
    if rank == 0:
        t0 = time.perf_counter()
        try:
            s.allreduce(np.ones(30))
            out["error"] = None
        except ts.TsdfError as e:
            out["error"] = str(e); out["code"] = e.code
        out["seconds"] = time.perf_counter() - t0
    else:
        time.sleep(8.0)                     # never takes part
print("RESULT " + json.dumps(out))
"""


def run_ranks(nranks, mode, tmp_path):
    name = f"/tsdf_peer_test_{os.getpid()}_{int(time.time() * 1e3) & 0xffffff}"
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT % {"root": ROOT})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(nranks), name, mode], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(nranks)]
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-3000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1][7:]))
    return sorted(outs, key=lambda o: o["rank"])


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_peer_exchange_sums_in_rank_order_on_every_rank(tmp_path, nranks):
    outs = run_ranks(nranks, "sums", tmp_path)
    for k in range(64):
        want = np.zeros(30)
        for r in range(nranks):               # rank order, starting from 0.0: the library's order, bit for bit
            want = want + np.array([(r + 1) * 0.1 + k + e * 1e-3 for e in range(30)])
        for o in outs:
            assert np.array_equal(np.array(o["rows"][k]), want), (k, o["rank"])
    for o in outs:
        assert o["after"] == [7.0] * 30


def test_a_rank_whose_peers_never_arrive_gives_up(tmp_path):
    outs = run_ranks(2, "absent", tmp_path)
    r0 = outs[0]
    assert r0["error"] and "peer exchange" in r0["error"], r0
    assert 4.0 < r0["seconds"] < 15.0           # the kernel's own 5 s limit, not a hung GPU (slack: a busy box)


def test_more_than_64_ranks_are_refused():
    import tracking_sdf_amd as ts
    s = ts.SDF(32)
    with pytest.raises(ts.TsdfError):
        s.comm_init_peer(65, 0, "/tsdf_peer_never")
