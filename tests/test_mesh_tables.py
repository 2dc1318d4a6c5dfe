"""Mesh extraction, CPU side: the generated marching-cubes case table, and known answers for the oracle's
restatement of pcl::MarchingCubesSDF::performReconstruction / SDF::interpolate_color
(reference src/marching_cubes_sdf.cpp:87-287, src/sdf.cpp:164-217, :353-383)."""
import os
import re
import subprocess
import sys
from collections import Counter

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_mc_tables as gen  # noqa: E402

REF_HEADER = "/root/reference/src/include/sdf_3d_reconstruction/marching_cubes_sdf.h"


def test_committed_tables_pass_the_structural_check():
    """Both committed headers identical, in the generator's layout, and every case tiles exactly the polygons derived
    from the cube geometry (needs no reference tree)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_mc_tables.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def committed_table():
    return gen.parse_header(open(os.path.join(ROOT, "tracking_sdf_amd", "csrc", "mc_tables.h")).read())


def test_table_structure():
    table = committed_table()
    derived = gen.build()
    assert table[0] == [] and table[255] == []
    for case in range(256):
        tris = table[case]
        used = {e for t in tris for e in t}
        mask = gen.edge_mask(case)
        assert used == {e for e in range(12) if mask >> e & 1}, case     # exactly the crossed edges
        loops = gen.loops_of(tris)
        assert loops is not None, case
        assert sum(len(lp) - 2 for lp in loops) == len(tris), case       # n - 2 triangles per polygon
        assert sorted(e for lp in loops for e in lp) == sorted(used), case   # every crossed edge on one polygon
        assert loops == gen.loops_of(derived[case]), case                # = the polygons the geometry dictates
        # complement = the same contour with the other side inside: same crossed edges
        assert gen.edge_mask(case ^ 255) == mask


@pytest.mark.skipif(not os.path.exists(REF_HEADER), reason="reference tree not present")
def test_identical_to_the_reference_table_in_all_256_cases():
    """tsdf_mesh_read must return performReconstruction's triangle soup bit for bit, so the committed table must BE the
    reference's: the same triangles in the same order in every case, and the same edge masks."""
    ref, masks = gen.parse_reference(REF_HEADER)
    table = committed_table()
    for case in range(256):
        assert masks[case] == gen.edge_mask(case), case
        assert table[case] == ref[case], case


# ---------------------------------------------------------------------------------------------------------

def _sphere(m=48, r=1.0, c=(0.0, 0.0, 1.25)):
    s = orc.SDF(m)
    s.create_circle(r, *c)
    return s


def weld(v, tol=1e-5):
    """Vertex ids of a triangle soup with vertices closer than tol merged: (n_tri, 3) ints."""
    from scipy.spatial import cKDTree
    pts = v.reshape(-1, 3).astype(np.float64)
    tree = cKDTree(pts)
    ids = np.arange(len(pts))
    for a, b in sorted(tree.query_pairs(tol)):
        ids[b] = ids[a] = min(ids[a], ids[b])
    for q in range(len(ids)):                      # path compression (pairs are sorted, chains are short)
        while ids[ids[q]] != ids[q]:
            ids[q] = ids[ids[q]]
    return ids.reshape(-1, 3).tolist()


def test_sphere_mesh_is_closed_and_on_the_surface():
    m = 48
    s = _sphere(m)
    v = s.mesh()
    assert len(v) > 1000
    # A vertex shared by two cubes is computed from either cube's corner positions (idx*extent/m vs
    # (idx-1)*extent/m + extent/m) and can differ in the last float bit, in the reference too: weld at 1e-5 m.
    ids = weld(v)
    cnt = Counter()
    for tri in ids:
        for a, b in ((0, 1), (1, 2), (2, 0)):
            cnt[(tri[a], tri[b])] += 1
    assert all(cnt.get((b, a), 0) == n for (a, b), n in cnt.items())     # every edge has its opposite: watertight
    # the reference's mesh frame is half a voxel off the voxel centres (marching_cubes_sdf.cpp:121-124)
    cell = np.array([6.0 / m, 6.0 / m, 3.5 / m])
    w = v.reshape(-1, 3) + np.array([-3.0, -3.0, -0.5]) + cell / 2
    rad = np.linalg.norm(w - np.array([0.0, 0.0, 1.25]), axis=1)
    assert rad.max() < 1.0 + 1e-5 and rad.min() > 1.0 - 0.5 * np.linalg.norm(cell) ** 2
    # winding: normals point to growing D (out of the sphere)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    ctr = v.mean(axis=1) + np.array([-3.0, -3.0, -0.5]) + cell / 2 - np.array([0.0, 0.0, 1.25])
    sgn = np.sign(np.einsum("ij,ij->i", n, ctr))
    assert abs(sgn.sum()) == len(v)                                       # one consistent orientation


def test_single_cube_known_answer():
    """One corner below the iso level: one triangle on edges 0, 8, 3 of that cube at the reference's positions."""
    m = 8
    s = orc.SDF(m, 8.0, 8.0, 4.0, (0.0, 0.0, 0.0))
    s.D[:] = 1.0
    s.W[:] = 1.0
    i, j, k = 3, 4, 2
    s.D[(i * m + j) * m + k] = -1.0
    v = s.mesh()
    # the voxel is corner 0 of cube (i,j,k) and a corner of 7 more cubes: 8 triangles around it
    assert v.shape == (8, 3, 3)
    cell = np.array([1.0, 1.0, 0.5], dtype=np.float32)
    base = np.array([i, j, k], dtype=np.float32) * cell
    # every vertex sits halfway to a neighbour (mu = (0 - (-1)) / (1 - (-1)) = 0.5)
    d = np.abs(v.reshape(-1, 3) - base) / cell
    assert np.allclose(np.sort(d, axis=1), [0.0, 0.0, 0.5])
    # the cube whose corner 0 is the voxel comes last in index order among ... check it explicitly:
    tri = [t for t in v if np.all(t >= base - 1e-6)]
    assert len(tri) == 1
    e0 = base + np.array([0.5, 0, 0]) * cell      # edge 0: corner 0 -> corner 1 (+x)
    e8 = base + np.array([0, 0.5, 0]) * cell      # edge 8: corner 0 -> corner 4 (+y)
    e3 = base + np.array([0, 0, 0.5]) * cell      # edge 3: corner 3 (+z) -> corner 0
    assert {tuple(p) for p in tri[0]} == {tuple(e0), tuple(e8), tuple(e3)}


def test_weight_gate_and_borders_and_iso_range():
    m = 10
    s = orc.SDF(m)
    s.create_circle(1.0, 0.0, 0.0, 1.25)
    full = len(s.mesh())
    assert full > 0
    # one unobserved voxel removes the 8 cubes that touch it and nothing else
    W_saved = s.W.copy()
    idx = np.flatnonzero((np.abs(s.D) < 0.2))[5]
    s.W[idx] = 0.0
    assert len(s.mesh()) < full
    s.W[:] = W_saved
    # boundary voxels are never cube bases (sdf.cpp:36-39): a surface that lives only in the first/last cube
    # layer of the grid is not meshed
    t = orc.SDF(m)
    t.W[:] = 1.0
    t.D[:] = 1.0
    t.D.reshape(m, m, m)[0, :, :] = -1.0          # sign change between layers 0 and 1: cubes with i = 0 only
    assert len(t.mesh()) == 0
    t.D.reshape(m, m, m)[:2, :, :] = -1.0         # between layers 1 and 2: cubes with i = 1
    assert len(t.mesh()) == 2 * (m - 2) * (m - 2)
    with pytest.raises(ValueError):
        s.mesh(iso_level=1.0)
    with pytest.raises(ValueError):
        s.mesh(iso_level=-0.1)
    assert len(s.mesh(iso_level=0.25)) > 0
    # the x-range restriction splits the soup without changing it
    a, b = s.mesh(i0=0, i1=5), s.mesh(i0=5, i1=m)
    assert np.array_equal(np.concatenate([a, b]), s.mesh())


def test_interpolate_color_known_answers():
    m = 8
    s = orc.SDF(m, 8.0, 8.0, 8.0, (0.0, 0.0, 0.0))
    R = s.R.reshape(m, m, m); G = s.G.reshape(m, m, m); B = s.B.reshape(m, m, m)
    R[:] = 100.0; G[:] = np.arange(m, dtype=np.float32)[None, :, None] * 10; B[:] = 51.0
    # no coloured corner: 0/0
    assert np.all(np.isnan(s.interpolate_color((2.0, 2.0, 2.0))[:3]))
    s.Color_W[:] = 1.0
    # exact voxel hit (world = centre of voxel (2,3,4)): stored values, NOT divided by 255 (sdf.cpp:195-200)
    c = s.interpolate_color((2.5, 3.5, 4.5))
    assert c.tolist() == [100.0, 30.0, 51.0, 1.0]
    # halfway between voxel (2,3,4) and (3,3,4): weights as in interpolate_distance (2,2,2/3 x4,2/5 x2), / 255
    c = s.interpolate_color((3.0, 3.5, 4.5))
    assert np.isclose(c[0], 100.0 / 255.0, rtol=1e-6) and np.isclose(c[2], 51.0 / 255.0, rtol=1e-6)
    wsum = 2 + 2 + 4 * (2.0 / 3.0) + 2 * 0.4
    g = (2 * 30 + 2 * 30 + (2.0 / 3.0) * (40 + 40 + 30 + 30) + 0.4 * (40 + 40)) / wsum / 255.0
    assert np.isclose(c[1], g, rtol=1e-6)
    # mesh colours: evaluated at vertex + origin
    s.D[:] = 1.0; s.W[:] = 1.0
    s.D.reshape(m, m, m)[:4] = -1.0
    v, col = s.mesh(with_color=True)
    assert col.shape == (len(v), 3, 4) and np.all(col[..., 3] == 1.0)
    k = 7
    assert np.array_equal(col.reshape(-1, 4)[k], s.interpolate_color(v.reshape(-1, 3)[k].astype(np.float64)))
