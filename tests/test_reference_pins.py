"""Pins against DATA the reference itself holds (tests/golden/*_ref.json, written by tools/pin_mc_tables.py from
/root/reference/cuda/includes/marching_cubes_constants.cu:3-275 and cuda/includes/bindings.h:5-75).

The reference ships no tests and no golden vectors; its marching-cubes tables and its FFI header are the only
reference-held data on the hot path.  Product and oracle share one table generator, so without this pin a table
error would be common-mode and invisible to HIP-vs-oracle parity."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_mc_tables as G  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _ref_tables():
    ref = json.load(open(os.path.join(GOLD, "mc_tables_ref.json")))
    rows = []
    for r in ref["triangle_table"]:
        assert len(r) == 16
        n = r.index(-1) if -1 in r else 16
        assert all(v == -1 for v in r[n:]), "entries after the terminator"
        rows.append(r[:n])
    return [tuple(e) for e in ref["edge_table"]], rows


def _decode_header(path):
    """(rows, counts, edge corner pairs, edge masks) decoded from a generated header's packed arrays."""
    txt = open(path).read()

    def arr(name):
        body = txt[txt.index(name):]
        body = body[body.index("{") + 1: body.index("}")]
        return [int(v.rstrip("ulUL"), 0) for v in re.findall(r"0x[0-9a-fA-F]+(?:ull)?|\d+", body)]

    packed, counts, corners, masks = arr("MC_TRI_PACKED"), arr("MC_TRI_COUNT"), arr("MC_EDGE_CORNERS"), arr("MC_EDGE_MASK")
    rows = []
    for w in packed:
        r = []
        for i in range(16):
            nib = (w >> (4 * i)) & 0xF
            if nib == 0xF:
                break
            r.append(nib)
        rows.append(r)
    return rows, counts, [(c & 0xF, c >> 4) for c in corners], masks


def test_generator_tables_equal_the_reference_tables():
    edges, rows = _ref_tables()
    assert [tuple(e) for e in G.EDGES] == edges
    assert [list(r) for r in G.TRI] == rows


def test_generated_headers_decode_to_the_reference_tables():
    edges, rows = _ref_tables()
    for rel in ("oracle/mc_tables.h", "bevy-signed-distance-mesh-generation_amd/csrc/mc_tables.hpp"):
        h_rows, h_counts, h_edges, h_masks = _decode_header(os.path.join(ROOT, rel))
        assert h_rows == rows, rel
        assert h_counts == [len(r) // 3 for r in rows], rel
        assert h_edges == edges, rel
        assert h_masks == [sum(1 << e for e in set(r)) for r in rows], rel


def test_edge_key_immediates_in_the_kernels_follow_the_reference_edge_table():
    """kernels.hip edge_key() packs, per cube edge, the sum of its two corners' offsets (0, 1 or 2 per axis) two bits
    each into three immediates; the corner offsets are compute_mesh_generation.cu:80-82 (+x iff c%4 in {1,2},
    +y iff c%4 >= 2, +z iff c >= 4)."""
    edges, _ = _ref_tables()
    cox = lambda c: 1 if c % 4 in (1, 2) else 0   # noqa: E731
    coy = lambda c: 1 if c % 4 >= 2 else 0        # noqa: E731
    coz = lambda c: 1 if c >= 4 else 0            # noqa: E731
    imm = [0, 0, 0]
    for e, (a, b) in enumerate(edges):
        for k, f in enumerate((cox, coy, coz)):
            imm[k] |= (f(a) + f(b)) << (2 * e)
    txt = open(os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd/csrc/kernels.hip")).read()
    m = re.search(r"ox = \((0x[0-9a-f]+)u >> \(2 \* e\)\) & 3u, oy = \((0x[0-9a-f]+)u >> \(2 \* e\)\) & 3u, oz = \((0x[0-9a-f]+)u", txt)
    assert m, "edge_key immediates not found"
    assert [int(m.group(i), 16) for i in (1, 2, 3)] == imm


def _our_header():
    txt = open(os.path.join(ROOT, "include/sdmesh/bindings.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return txt


def test_bindings_constants_equal_the_reference_header():
    ref = json.load(open(os.path.join(GOLD, "bindings_ref.json")))["constants"]
    txt = _our_header()
    for name in ("BLOCK_SIZE", "MESH_GENERATION_INIT_FACTOR", "MESH_GENERATION_BB_SIZE"):
        m = re.search(r"#define\s+%s\s+(\S+)" % name, txt)
        assert m and m.group(1) == ref[name], name


def test_bindings_struct_fields_equal_the_reference_header():
    """Same struct names, field names, C types, array extents and declaration order (=> same layout under the C ABI;
    the numeric sizes/offsets are static-asserted in the header and checked in tests/test_abi.py)."""
    ref = json.load(open(os.path.join(GOLD, "bindings_ref.json")))["structs"]
    txt = _our_header()
    ours = {}
    for m in re.finditer(r"struct\s+(\w+)\s*\{(.*?)\}\s*;", txt, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?)\s*(\**)\s*(\w+)\s*(\[\s*\d+\s*\])?$", decl)
            ctype = (fm.group(1) + (" " + fm.group(2) if fm.group(2) else "")).strip()
            fields.append({"name": fm.group(3), "type": ctype, "array": int(re.sub(r"\D", "", fm.group(4))) if fm.group(4) else 0})
        ours[m.group(1)] = fields
    assert set(ref) <= set(ours), set(ref) - set(ours)
    for name, fields in ref.items():
        assert ours[name] == fields, name


def test_path_constants_of_oracle_and_product_equal_the_reference_text():
    """The numeric literals of the hot path as the reference's own source text holds them (tests/golden/path_constants_ref.json,
    parsed by tools/pin_mc_tables.py: numbers, not code) against the oracle's C restatement AND the product's device code -- the two
    are written independently of each other here, but both from the survey, so a misread constant would otherwise be common-mode."""
    ref = json.load(open(os.path.join(GOLD, "path_constants_ref.json")))
    oracle = open(os.path.join(ROOT, "oracle", "sdmesh_oracle.c")).read()
    sdf = open(os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "sdf.hpp")).read()
    kern = open(os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "kernels.hip")).read()
    meshdev = open(os.path.join(ROOT, "bevy-signed-distance-mesh-generation_amd", "csrc", "mesh_device.hpp")).read()
    f = lambda s: float(s)                                                       # noqa: E731
    # finite-difference stencil: step, 8x inner weight, +-2 eps outer offsets
    eps = f(ref["normal_epsilon"])
    assert f(re.search(r"#define NORMAL_EPSILON ([0-9.]+)f", oracle).group(1)) == eps
    assert f(re.search(r"#define SDM_NORMAL_EPS ([0-9.]+)f", sdf).group(1)) == eps
    assert ref["stencil_weights"] == ["8.0"] and ref["stencil_offsets_in_epsilons"] == ["-2.0", "2.0"]
    assert "8.0f * F(NORMAL_EPSILON, 0.0f, 0.0f)" in oracle and "F(2.0f * NORMAL_EPSILON, 0.0f, 0.0f)" in oracle
    assert "e2 = 2.0f * SDM_NORMAL_EPS" in sdf and re.search(r"8\.0f \* \w+\.x - 8\.0f \* \w+\.y", sdf)
    # Newton projection: threshold (inclusive), iteration cap
    thr, cap = f(ref["newton_threshold"]), int(ref["newton_iteration_cap"])
    m = re.search(r"!collision && i < (\d+); i\+\+\) \{.*?collision = gabs\(sd\) <= ([0-9.]+)f;", oracle, flags=re.S)
    assert int(m.group(1)) == cap and f(m.group(2)) == thr
    assert int(re.search(r"#define SDM_MAX_NEWTON (\d+)", sdf).group(1)) == cap
    got = {f(v) for v in re.findall(r"fabsf\(sd\) <= ([0-9.]+)f", sdf + kern)}
    assert got == {thr}, got
    # marching cubes: midpoint, `<= 0` inside tests, 5 triangle slots per voxel, flip when the dot product is `<= 0`, centroid / 3
    assert f(ref["mc_midpoint_factor"]) == 0.5 and ref["mc_inside_test"] == "<=" and ref["refine_inside_test"] == "<="
    assert ref["orientation_flip_test"] == "<=" and f(ref["centroid_divisor"]) == 3.0 and int(ref["triangles_per_voxel_slot"]) == 5
    assert re.search(r"<= 0\.0f\) bits \|=", kern) and "sdf_eval(scene, params, " in oracle and re.search(r"<= 0\.0f", oracle)
    assert "a / 3.0f" in sdf and "vdivs(vadd(vadd(v0, v1), v2), 3.0f)" in oracle and "dot3(tn, an) <= 0.0f" in oracle
    assert re.search(r"5u? \* ", meshdev + kern) and re.search(r"5 \* ", oracle)
    # sd_obj: skeleton (3, 1, 0.5) with line width 0.1, unit sphere, smooth minimum k = 0.5 with h^3 k / 6
    sk = [f(v) for v in ref["sd_obj"]["skeleton_size"]]
    assert sk == [3.0, 1.0, 0.5] and f(ref["sd_obj"]["skeleton_line_width"]) == 0.1 and f(ref["sd_obj"]["sphere_radius"]) == 1.0
    assert f(ref["sd_obj"]["smooth_k"]) == 0.5 and ref["smooth_min"]["sixth"] == ["1.0", "6.0"]
    assert "sd_box_skeleton(p, 0, (3, 1, 0.5), 0.1)" in sdf and "h * h * h * k * (1.0f / 6.0f)" in sdf and "h * h * h * k * (1.0f / 6.0f)" in oracle
    # (the product evaluates smooth_min(a1, a2, 0.5f) as smooth_min_half: h^3 * (fl(1/6) / 2), the same value with one multiplication fewer)
    assert "smooth_min_half(a1, a2)" in sdf and "fmax1(0.5f - fabsf(a - b), 0.0f) / 0.5f" in sdf
    assert np.float32(float(re.search(r"#define SDM_SIXTH_HALF ([0-9.]+)f", sdf).group(1))) == np.float32(1.0) / np.float32(6.0) / np.float32(2.0)
    assert re.search(r"- 0\.1f", sdf) and re.search(r"\) - 1\.0f;", sdf)
    assert re.search(r"3\.0f,\s*1\.0f,\s*0\.5f", oracle) and re.search(r"0\.1f", oracle)
    # sd_mandelbulb (the field scene): power, time factor, iteration count, escape radius, the estimate's factor -- in oracle and product
    mb = ref["mandelbulb"]
    assert (mb["power"], mb["time_factor"], mb["iterations"], mb["escape_radius"], mb["estimate_factor"], mb["unit_scale"]) == ("7.0", "0.001", "25", "2.0", "0.5", "0.4")
    for txt, fn in ((oracle, "static float sd_mandelbulb("), (sdf, "SDM_HD float sd_mandelbulb(")):
        body = txt[txt.index(fn):]
        body = body[:body.index("\n}\n")]
        assert re.search(r"power = %sf \* \(1\.0f \+ time \* %sf\)" % (re.escape(mb["power"]), re.escape(mb["time_factor"])), body), fn
        assert re.search(r"i < %s; i\+\+" % mb["iterations"], body) and re.search(r"if \(r > %sf\) break;" % re.escape(mb["escape_radius"]), body), fn
        assert re.search(r"return %sf \* \w+\(r\) \* r / dr;" % re.escape(mb["estimate_factor"]), body), fn
    # weld: key = round(x * 10e4) per coordinate, triangles dropped on a non-finite x of their first vertex
    q = f(ref["weld_quantum_factor"])
    assert q == 1e5 and ref["weld_finite_test"] == "x"
    assert f(re.search(r"roundf\(x \* ([0-9.e]+)f\)", oracle).group(1)) == q and f(re.search(r"roundf\(x \* ([0-9.e]+)f\)", sdf).group(1)) == q
