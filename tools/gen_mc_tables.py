#!/usr/bin/env python3
"""The marching-cubes case table used by the mesh-extraction kernels and by the oracle.

The reference meshes the volume with a PCL-style marching cubes driven by the classic 256-case edge / triangle table
(the Lorensen-Cline cases as tabulated by Cory Bloyd / Paul Bourke; in the reference:
src/include/sdf_3d_reconstruction/marching_cubes_sdf.h:73-364, consumed at src/marching_cubes_sdf.cpp:100-199).
`tsdf_mesh_read` is specified to return the triangle soup of performReconstruction bit for bit, and inside a cube
that soup is decided by the table's choice of diagonals and by its triangle order -- so the committed headers hold
that table's constants, as data, in this repository's own layout (kMcNumTri / kMcTri).

Round 1 shipped a table DERIVED from the cube geometry by this script (same polygons in all 256 cases, other
diagonals in 158 of them); the derivation is kept as the structural check of the committed constants:

  * corner c sits at the reference's offsets (marching_cubes_sdf.cpp:129-141): bit test of c gives
    0:(0,0,0) 1:(x) 2:(x,z) 3:(z) 4:(y) 5:(x,y) 6:(x,y,z) 7:(y,z);
  * edge e joins the corner pairs of marching_cubes_sdf.cpp:146-169 (e0 = 0-1 ... e11 = 3-7);
  * a corner is "inside" when its value is below the iso level (bit c of the case number, :108-115);
  * on every cube face the crossed edges are joined into contour segments oriented with the inside on their
    left seen from outside the cube; a face with four crossings (two diagonal inside corners) is ambiguous and
    is resolved by cutting off each inside corner -- one rule for every cube, so two cubes sharing an ambiguous
    face draw the same two segments on it (no cracks);
  * segments chain into closed loops (every crossed edge ends one segment and starts another).

--check (run by tests/test_mesh_tables.py, needs no reference tree): the two committed headers are identical, and in
every one of the 256 cases the committed triangles use exactly the crossed edges and tile exactly the polygons the
derivation above yields, with the derivation's winding.
--from-reference HEADER (build container only): read the table out of the reference header and write the two
committed headers; refuses to write if the structural check fails.

Usage:  python tools/gen_mc_tables.py --check
        python tools/gen_mc_tables.py --from-reference /root/reference/src/include/sdf_3d_reconstruction/marching_cubes_sdf.h
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


def loops_of(tris):
    """Boundary cycles of a set of oriented triangles over edge numbers (None if not a set of closed polygons)."""
    d = {}
    for t in tris:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            if (b, a) in d:
                del d[(b, a)]
            else:
                d[(a, b)] = 1
    nxt = {}
    for a, b in d:
        if a in nxt:
            return None
        nxt[a] = b
    out, seen = [], set()
    for s in sorted(nxt):
        if s in seen:
            continue
        loop, c = [], s
        while c not in seen:
            seen.add(c)
            loop.append(c)
            c = nxt.get(c)
            if c is None:
                return None
        out.append(tuple(loop))       # starts at its smallest edge because of the sorted() walk
    return sorted(out)


def structural_errors(table):
    """Cases in which `table` does not tile the derived polygons (same crossed edges, same loops, same winding)."""
    derived = build()
    bad = []
    for case in range(256):
        tris = table[case]
        used = {e for t in tris for e in t}
        mask = edge_mask(case)
        ok = used == {e for e in range(12) if mask >> e & 1} and len(tris) == len(derived[case]) and len(tris) <= 5
        ok = ok and loops_of(tris) == loops_of(derived[case])
        if not ok:
            bad.append(case)
    return bad


def parse_reference(path):
    """The 256 triangle lists of the reference header's triTable (and its edgeTable, for the mask check)."""
    import re
    src = open(path).read()
    body = src[src.index("{", src.index("triTable")):]
    table = []
    for r in re.findall(r"\{([^{}]*)\}", body)[:256]:
        v = [int(x) for x in r.replace("\n", " ").split(",") if x.strip()]
        v = v[:v.index(-1)] if -1 in v else v
        table.append([tuple(v[k:k + 3]) for k in range(0, len(v), 3)])
    eb = src[src.index("{", src.index("edgeTable")):]
    masks = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", eb[:eb.index("}")])]
    assert len(table) == 256 and len(masks) == 256
    return table, masks


def parse_header(text):
    """kMcTri of a committed mc_tables.h -> 256 triangle lists."""
    import re
    body = text[text.index("kMcTri[256][16]"):]
    rows = re.findall(r"\{([^{}]*)\}", body)[:256]
    table = []
    for r in rows:
        v = [int(x) for x in r.split(",") if x.strip()]
        assert len(v) == 16
        v = v[:v.index(-1)] if -1 in v else v
        table.append([tuple(v[k:k + 3]) for k in range(0, len(v), 3)])
    counts = [int(x) for x in re.findall(r"\d+", text[text.index("kMcNumTri[256]") + 14:text.index("};")])]
    assert len(table) == 256 and counts == [len(t) for t in table], "kMcNumTri does not match kMcTri"
    return table


def render(table):
    lines = [
        "// mc_tables.h -- written by tools/gen_mc_tables.py --from-reference; do not edit.",
        "// The classic 256-case marching-cubes triangle table (Lorensen-Cline cases as tabulated by Bloyd / Bourke), the",
        "// constants of the reference's marching_cubes_sdf.h:73-364, in this repository's layout: kMcNumTri[case] triangles,",
        "// kMcTri[case][3*t+{0,1,2}] = edge numbers (corner / edge numbering of marching_cubes_sdf.cpp:108-169).  Kept as",
        "// data because tsdf_mesh_read must return performReconstruction's triangle soup (same diagonals, same order);",
        "// tools/gen_mc_tables.py --check verifies every case against the polygons derived from the cube geometry.",
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


HEADERS = ("tracking_sdf_amd/csrc/mc_tables.h", "oracle/mc_tables.h")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="verify the committed headers (structure; both copies identical)")
    ap.add_argument("--from-reference", metavar="HEADER", help="write the committed headers from the reference's table")
    args = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if args.from_reference:
        table, masks = parse_reference(args.from_reference)
        assert masks == [edge_mask(c) for c in range(256)], "edgeTable disagrees with the corner / edge numbering"
        bad = structural_errors(table)
        if bad:
            print("table does not tile the derived polygons in cases", bad)
            return 1
        text = render(table)
        for rel in HEADERS:
            with open(os.path.join(root, rel), "w") as f:
                f.write(text)
            print("wrote", rel)
        return 0
    texts = [open(os.path.join(root, rel)).read() for rel in HEADERS]
    rc = 0
    if texts[0] != texts[1]:
        print("the two committed headers differ")
        rc = 1
    table = parse_header(texts[0])
    if render(table) != texts[0]:
        print("committed header is not in the generator's layout")
        rc = 1
    bad = structural_errors(table)
    if bad:
        print("cases that do not tile the derived polygons:", bad)
        rc = 1
    if not args.check:
        print("nothing written: use --from-reference HEADER (or --check)")
    return rc


if __name__ == "__main__":
    sys.exit(main())
