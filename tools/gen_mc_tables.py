#!/usr/bin/env python3
"""Generate the marching-cubes case table used by the mesh-extraction kernels and by the oracle.

The reference meshes the volume with a PCL-style marching cubes whose 256-case triangle table is Paul
Bourke's hand-made one (src/include/sdf_3d_reconstruction/marching_cubes_sdf.h:73-364).  That table is data
from a third party and is not reproduced here.  This script DERIVES a table from the cube's geometry:

  * corner c sits at the reference's offsets (marching_cubes_sdf.cpp:129-141): bit test of c gives
    0:(0,0,0) 1:(x) 2:(x,z) 3:(z) 4:(y) 5:(x,y) 6:(x,y,z) 7:(y,z);
  * edge e joins the corner pairs of marching_cubes_sdf.cpp:146-169 (e0 = 0-1 ... e11 = 3-7);
  * a corner is "inside" when its value is below the iso level (bit c of the case number, :108-115);
  * on every cube face the crossed edges are joined into contour segments oriented with the inside on their
    left seen from outside the cube; a face with four crossings (two diagonal inside corners) is ambiguous and
    is resolved by cutting off each inside corner -- one rule for every cube, so two cubes sharing an ambiguous
    face draw the same two segments on it (no cracks);
  * segments chain into closed loops (every crossed edge ends one segment and starts another);
  * each loop is rotated to start at its smallest edge number, loops are ordered by that number and
    fan-triangulated.

So the *polygons* of a case are determined by the geometry and the ambiguity rule; only the choice of diagonals
and the order of the triangles are this script's own.  tests/test_mesh_tables.py checks the structural
properties (every crossed edge used once per loop, closed loops, triangle counts) and, when the reference tree
is present, that all 256 cases have exactly the reference table's polygons (same loops, same winding) and the
same edge masks -- i.e. the mesh has the same vertices and the same number of triangles per cube; 98 cases
are the identical triangle set, in the others a polygon with more than three vertices is cut along other
diagonals.

Usage:  python tools/gen_mc_tables.py [--check]     (writes tracking_sdf_amd/csrc/mc_tables.h, oracle/mc_tables.h)
"""
import argparse
import os
import sys

CORNER = [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 1, 1), (0, 1, 1)]
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
EDGE_ID = {}
for _e, (_a, _b) in enumerate(EDGE):
    EDGE_ID[(_a, _b)] = _e
    EDGE_ID[(_b, _a)] = _e


def _faces():
    """The six faces as corner cycles, counter-clockwise seen from outside the cube."""
    out = []
    for axis in range(3):
        for side in (0, 1):
            cs = [c for c in range(8) if CORNER[c][axis] == side]
            # order the four corners cyclically: neighbours differ in exactly one coordinate
            cyc = [cs[0]]
            left = cs[1:]
            while left:
                for c in left:
                    if sum(abs(CORNER[c][d] - CORNER[cyc[-1]][d]) for d in range(3)) == 1:
                        cyc.append(c)
                        left.remove(c)
                        break
            a, b, c = (CORNER[cyc[0]], CORNER[cyc[1]], CORNER[cyc[2]])
            u = [b[d] - a[d] for d in range(3)]
            v = [c[d] - b[d] for d in range(3)]
            n = [u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]]
            outward = 1 if side == 1 else -1
            if n[axis] * outward < 0:
                cyc.reverse()
            out.append(cyc)
    return out


FACES = _faces()


def separate_inside(case):
    """Ambiguity rule: True = cut off the inside corners of an ambiguous face, False = the outside corners."""
    return True


def case_loops(case, rule=separate_inside):
    """Closed loops of edge numbers for one case, inside on the left seen from outside."""
    inside = [(case >> c) & 1 for c in range(8)]
    nxt = {}
    for cyc in FACES:
        n_cross = 0
        for q in range(4):
            a, b = cyc[q], cyc[(q + 1) % 4]
            if inside[a] != inside[b]:
                n_cross += 1
        if n_cross == 0:
            continue
        if n_cross == 2:
            L = E = None
            for q in range(4):
                a, b = cyc[q], cyc[(q + 1) % 4]
                if inside[a] and not inside[b]:
                    L = EDGE_ID[(a, b)]
                if not inside[a] and inside[b]:
                    E = EDGE_ID[(a, b)]
            nxt[L] = E
            continue
        # four crossings
        sep_in = rule(case)
        for q in range(4):
            c, prv, nx = cyc[q], cyc[(q - 1) % 4], cyc[(q + 1) % 4]
            if sep_in and inside[c]:
                nxt[EDGE_ID[(c, nx)]] = EDGE_ID[(prv, c)]       # L on c->next, E on prev->c
            if not sep_in and not inside[c]:
                nxt[EDGE_ID[(prv, c)]] = EDGE_ID[(c, nx)]       # L on prev->c, E on c->next
    loops, seen = [], set()
    for e in sorted(nxt):
        if e in seen:
            continue
        loop, cur = [], e
        while cur not in seen:
            seen.add(cur)
            loop.append(cur)
            cur = nxt[cur]
        assert cur == e, "open contour"
        loops.append(loop)
    return loops


def case_triangles(case, rule=separate_inside, flip=True):
    tris = []
    for loop in case_loops(case, rule):
        k = loop.index(min(loop))
        loop = loop[k:] + loop[:k]
        for q in range(1, len(loop) - 1):
            t = (loop[0], loop[q], loop[q + 1])
            tris.append((t[0], t[2], t[1]) if flip else t)
    return tris


def build():
    table = [case_triangles(c) for c in range(256)]
    assert max(len(t) for t in table) <= 5
    return table


def edge_mask(case):
    m = 0
    for e, (a, b) in enumerate(EDGE):
        if ((case >> a) & 1) != ((case >> b) & 1):
            m |= 1 << e
    return m


def render(table):
    lines = [
        "// mc_tables.h -- GENERATED by tools/gen_mc_tables.py; do not edit.",
        "// Marching-cubes case table derived from the cube geometry (corner / edge numbering of the reference's",
        "// marching_cubes_sdf.cpp:108-169): kMcNumTri[case] triangles, kMcTri[case][3*t+{0,1,2}] = edge numbers.",
        "// Not Bourke's table: the same polygons in every case (tests/test_mesh_tables.py), own diagonals and order.",
        "#pragma once",
        "",
        "#ifndef MC_TABLE_DECL",
        "#define MC_TABLE_DECL static const",
        "#endif",
        "",
        "MC_TABLE_DECL unsigned char kMcNumTri[256] = {",
    ]
    for r in range(0, 256, 32):
        lines.append("    " + ", ".join(str(len(table[c])) for c in range(r, r + 32)) + ",")
    lines.append("};")
    lines.append("")
    lines.append("MC_TABLE_DECL signed char kMcTri[256][16] = {")
    for c in range(256):
        flat = [e for t in table[c] for e in t]
        flat += [-1] * (16 - len(flat))
        lines.append("    {" + ", ".join("%2d" % v for v in flat) + "},   // %3d  edges 0x%03x" % (c, edge_mask(c)))
    lines.append("};")
    lines.append("")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="fail if the committed headers differ from the generator")
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = render(build())
    rc = 0
    for rel in ("tracking_sdf_amd/csrc/mc_tables.h", "oracle/mc_tables.h"):
        path = os.path.join(root, rel)
        if args.check:
            if not os.path.exists(path) or open(path).read() != text:
                print("stale:", rel)
                rc = 1
        else:
            with open(path, "w") as f:
                f.write(text)
            print("wrote", rel)
    return rc


if __name__ == "__main__":
    sys.exit(main())
