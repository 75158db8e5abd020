#!/usr/bin/env python3
"""Pin the data the reference itself holds for the hot path into tests/golden/ (build container only).

The reference ships no tests, so the only reference-held DATA on the path is
  * the marching-cubes tables   cuda/includes/marching_cubes_constants.cu:3-16 (12 edge corner pairs),
                                :18-275 (256 rows of 16 edge ids, -1 terminated)
  * the three FFI constants     cuda/includes/bindings.h:7-10  (BLOCK_SIZE, MESH_GENERATION_INIT_FACTOR,
                                MESH_GENERATION_BB_SIZE)
  * the FFI struct field lists  cuda/includes/bindings.h:43-75 (names and C types in declaration order)
  * the numeric literals of the path (stencil step and weights, Newton threshold and cap, midpoint factor, sd_obj's shape parameters,
                                the weld quantum ...)  signed_distance.cu, common.cu, marching_cubes.cu, compute_mesh_generation.cu, src/cuda/mod.rs
This script parses those files as TEXT (no compilation, nothing executed) and writes the values -- a constant
table and a handful of numbers, not source -- to tests/golden/mc_tables_ref.json, bindings_ref.json and path_constants_ref.json.
tests/test_reference_pins.py then checks tools/gen_mc_tables.TRI/EDGES, oracle/mc_tables.h,
csrc/mc_tables.hpp and include/sdmesh/bindings.h against them on every run (the reference itself does not travel).

Usage: python tools/pin_mc_tables.py [/root/reference]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return re.sub(r"//[^\n]*", " ", txt)


def braces_after(txt, name):
    """The balanced {...} initialiser that follows `name` in txt."""
    i = txt.index(name)
    i = txt.index("{", i)
    depth, j = 0, i
    while True:
        if txt[j] == "{":
            depth += 1
        elif txt[j] == "}":
            depth -= 1
            if depth == 0:
                return txt[i:j + 1]
        j += 1


def int_rows(block):
    inner = block.strip()[1:-1]
    return [[int(v) for v in re.findall(r"-?\d+", row)] for row in re.findall(r"\{([^{}]*)\}", inner)]


def pin_tables(ref):
    path = os.path.join(ref, "cuda/includes/marching_cubes_constants.cu")
    txt = strip_comments(open(path).read())
    edges = int_rows(braces_after(txt, "MC_EDGE_TABLE"))
    rows = int_rows(braces_after(txt, "MC_TRIANGLE_TABLE"))
    assert len(edges) == 12 and all(len(e) == 2 for e in edges), "edge table shape"
    assert len(rows) == 256 and all(len(r) == 16 for r in rows), "triangle table shape"
    return {"source": "cuda/includes/marching_cubes_constants.cu:3-16 (MC_EDGE_TABLE), :18-275 (MC_TRIANGLE_TABLE)",
            "edge_table": edges, "triangle_table": rows}


def pin_bindings(ref):
    path = os.path.join(ref, "cuda/includes/bindings.h")
    txt = strip_comments(open(path).read())
    consts = {}
    for name in ("BLOCK_SIZE", "MESH_GENERATION_INIT_FACTOR", "MESH_GENERATION_BB_SIZE", "CUDA_DEBUG"):
        m = re.search(r"#define\s+%s\s+(\S+)" % name, txt)
        if m:
            consts[name] = m.group(1)
    structs = {}
    for m in re.finditer(r"struct\s+(\w+)\s*\{(.*?)\}\s*;", txt, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fm = re.match(r"(.+?)\s*(\**)\s*(\w+)\s*(\[\s*\d+\s*\])?$", decl)
            ctype = (fm.group(1) + (" " + fm.group(2) if fm.group(2) else "")).strip()
            fields.append({"name": fm.group(3), "type": ctype, "array": int(re.sub(r"\D", "", fm.group(4))) if fm.group(4) else 0})
        structs[m.group(1)] = fields
    return {"source": "cuda/includes/bindings.h:5-10 (constants), :12-75 (structs, declaration order)",
            "constants": consts, "structs": structs}


def pin_path_constants(ref):
    """The numeric literals of the hot path as the reference's source text holds them (numbers, not code): the stencil step and
    weights, the Newton loop's threshold and cap, the midpoint factor, sd_obj's shape parameters, the weld quantum, the launch
    block size used by the host, the level-0 grid loop."""
    def text(rel):
        return strip_comments(open(os.path.join(ref, rel)).read())

    def one(pattern, txt, what):
        m = re.search(pattern, txt, flags=re.S)
        assert m, "reference text changed: " + what
        return m.groups() if m.lastindex and m.lastindex > 1 else m.group(1)

    sd = text("cuda/includes/signed_distance.cu")
    common = text("cuda/modules/common.cu")
    mc = text("cuda/includes/marching_cubes.cu")
    gen = text("cuda/modules/compute_mesh_generation.cu")
    host = open(os.path.join(ref, "src/cuda/mod.rs")).read()
    out = {"source": "cuda/includes/signed_distance.cu:20-57,177-240; cuda/modules/common.cu:222-226; cuda/includes/marching_cubes.cu:13-21; "
                     "cuda/modules/compute_mesh_generation.cu:12-120; src/cuda/mod.rs:105-122,163-174,268"}
    out["normal_epsilon"] = one(r"#define\s+NORMAL_EPSILON\s+([0-9.eE+-]+)f", sd, "NORMAL_EPSILON")
    normal = sd[sd.index("empirical_normal("):sd.index("struct NormalPlane")]
    out["stencil_weights"] = sorted(set(re.findall(r"([0-9.]+)f\s*\*\s*sd_func", normal)))
    out["stencil_offsets_in_epsilons"] = sorted(set(re.findall(r"(-?[0-9.]+)f\s*\*\s*NORMAL_EPSILON", normal)))
    newton = sd[sd.index("closest_surface_point("):]
    out["newton_iteration_cap"] = one(r"i\s*<\s*(\d+)", newton, "iteration cap")
    out["newton_threshold"] = one(r"abs\(sd\)\s*<=\s*([0-9.eE+-]+)f", newton, "collision threshold")
    out["smooth_min"] = {"sixth": one(r"h \* h \* h \* k \* \(([0-9.]+)f\s*/\s*([0-9.]+)f\)", sd, "smooth_min")}
    obj = common[common.index("float sd_obj("):]
    obj = obj[:obj.index("}")]
    out["sd_obj"] = {"skeleton_size": list(one(r"sd_box_skeleton\(p,\s*vec3\(0\.0f\),\s*vec3\(([0-9.]+)f,\s*([0-9.]+)f,\s*([0-9.]+)f\)", obj, "skeleton size")),
                     "skeleton_line_width": one(r"vec3\([0-9.f, ]+\),\s*([0-9.]+)f\)", obj, "line width"),
                     "sphere_radius": one(r"length\(p\)\s*-\s*([0-9.]+)f", obj, "sphere radius"),
                     "smooth_k": one(r"smooth_min\(a1,\s*a2,\s*([0-9.]+)f\)", obj, "smooth k")}
    bulb = sd[sd.index("float sd_mandelbulb("):sd.index("float sd_unit_mandelbulb(")]
    out["mandelbulb"] = {"power": one(r"#define\s+POWER\s+([0-9.]+)f", sd, "POWER"),
                         "time_factor": one(r"time\s*\*\s*([0-9.]+)f", bulb, "time factor"),
                         "iterations": one(r"i\s*<\s*(\d+)", bulb, "iterations"),
                         "escape_radius": one(r"r\s*>\s*([0-9.]+)f", bulb, "escape radius"),
                         "estimate_factor": one(r"return\s+([0-9.]+)f\s*\*\s*log\(r\)", bulb, "estimate factor"),
                         "unit_scale": one(r"sd_mandelbulb\(p\s*/\s*([0-9.]+)f", sd, "unit scale")}
    out["mc_midpoint_factor"] = one(r"float v = ([0-9.]+)f;", mc, "midpoint factor")
    out["mc_inside_test"] = one(r"cube\.values\[i\]\s*(<=|<)\s*0\.0f", mc, "inside test")
    out["refine_inside_test"] = one(r"sd_obj\([^)]*\)\s*(<=|<)\s*0\.0f", gen, "refine inside test")
    out["triangles_per_voxel_slot"] = one(r"triangle_start\s*=\s*(\d+)\s*\*\s*id", gen, "slots per voxel")
    out["orientation_flip_test"] = one(r"dot\(triangle_normal,\s*actual_normal\)\s*(<=|<)\s*0\.0f", gen, "flip test")
    out["centroid_divisor"] = one(r"\(v0 \+ v1 \+ v2\)\s*/\s*([0-9.]+)f", gen, "centroid")
    out["weld_quantum_factor"] = one(r"\(x \* ([0-9.eE+-]+)f32\)\.round\(\) as i64", host, "weld quantum")
    out["weld_finite_test"] = one(r"triangle\.vertices\[0\]\.position\.(\w)\.is_finite\(\)", host, "finite test")
    return out


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(out, "mc_tables_ref.json"), "w") as f:
        json.dump(pin_tables(ref), f, separators=(",", ":"))
        f.write("\n")
    with open(os.path.join(out, "bindings_ref.json"), "w") as f:
        json.dump(pin_bindings(ref), f, indent=1)
        f.write("\n")
    with open(os.path.join(out, "path_constants_ref.json"), "w") as f:
        json.dump(pin_path_constants(ref), f, indent=1)
        f.write("\n")
    print("pinned", out)


if __name__ == "__main__":
    main()
