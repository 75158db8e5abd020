"""The driver's contract for bench.py, checked on the GPU box: ONE JSON line on stdout with the required keys, for the N = 1 run
and for the degenerate one-rank form of the N > 1 run (`--one-mesh-selftest`: the same code path torch.distributed.run drives)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config")


def _run(*flags):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                    # exactly one line on stdout (RCCL's banner etc. go to stderr)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_single_gpu_line_keeps_the_contract():
    j = _run("--steps", "20", "--warmup", "3", "--no-sizes", "--no-anim", "--no-pipelined")
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 3 and j["higher_is_better"] is True
    assert j["unit"] == "Mvoxels/s" and j["dtype"] == "f32" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 512 ** 3 / (j["ms_per_step"] * 1e-3) / 1e6) / j["value"] < 1e-3
    assert (j["triangles"], j["vertices"]) == (530536, 265262)      # SURVEY 8(c), sd_obj k = 4
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - 4 * 513 ** 3 / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    assert r["traffic"] is None or abs(r["traffic"] - 4 * 513 ** 3) / (4 * 513 ** 3) < 0.05
    # the bare store stream beside the pass that computes: within a tenth of it either way (by box the fill is 4 % faster or 3 % slower than
    # the sphere pass, whose arithmetic paces its stores: MEASUREMENTS.md "Round 4, last")
    assert r["write_ceiling"]["achieved"] > 0.9 * r["achieved"]
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    s = j["roofline_step"]
    assert s["project"]["wave_instr"] > 0 and 0.0 < s["project"]["frac"] < 1.0
    assert s["project"].get("frac_of_issue_bound") is None or 0.0 < s["project"]["frac_of_issue_bound"] <= 1.0
    assert set(s["hbm_passes"]) >= {"P3_emit_active_list", "P4_triangles", "P5_weld"}
    for sc in ("sphere", "sdobj"):
        assert set(j["roofline_by_grid"][sc]) == {"256", "512", "1024"} and all(0.0 < r["frac"] < 1.0 for r in j["roofline_by_grid"][sc].values())
    ra = j["reference_abi_kernels"]
    assert ra["voxels_meshed"] == 265256 and ra["kernels_total_ms"] > ra["native_chain_device_ms"] > 0 and len(ra["per_level"]) == 4


@pytest.mark.gpu
def test_one_mesh_line_keeps_the_contract_on_one_rank():
    j = _run("--one-mesh-selftest", "--steps", "8", "--warmup", "2", "--max-levels", "5", "--no-replicas")
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 1 and j["scaling"] == "strong" and j["ranks_seen"] == 1 and j["one_mesh"]["comm_selftest_mismatches"] == 0
    assert j["config"]["one_mesh_form"] in ("zslab_indexed_gather", "xslab_distributed_weld")
    for form in ("zslab_indexed_gather", "xslab_distributed_weld"):
        for size in ("256^3", "512^3", "1024^3"):
            row = j["one_mesh"][form][size]
            assert row["t_1rank_ms"] > 0 and row["ms_per_regen"] > 0 and row["triangles"] > 0
    assert (j["one_mesh"]["xslab_distributed_weld"]["512^3"]["triangles"], j["one_mesh"]["xslab_distributed_weld"]["512^3"]["vertices"]) == (530536, 265262)
    d = j["dense_p1_sharded"]
    for size in ("256^3", "512^3", "1024^3"):
        for sc in ("sphere", "sdobj"):
            row = d[size][sc]
            assert row["peak"] == 8000.0 and 0.0 < row["frac"] < 1.0 and row["t_1rank_ms"] > 0 and row["ms_per_pass"] > 0
    assert d["512^3"]["sphere"]["planes_of_rank0"] == [0, 513]


@pytest.mark.gpu
def test_the_multi_rank_bench_path_runs_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nproc-per-node 3 bench.py --gpus 3 ...` exactly as the driver starts it, except that all three ranks use
    GPU 0 (`--rehearse-on-one-gpu`: process group gloo, the library's collectives host-staged instead of RCCL, which refuses duplicate
    devices).  Every line of the N > 1 path runs -- slab balancing, sdmesh_sharded_regen and sdmesh_distributed_regen between real
    processes, the count checks against the 1-GPU mesh, the dense P1 slabs, the replicas, the one JSON line on rank 0 -- and the line says
    that its timings are not a measurement."""
    port = 29800 + os.getpid() % 150
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "6", "--warmup", "2",
                        "--rehearse-on-one-gpu", "--max-levels", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    for k in REQUIRED:
        assert k in j, k
    assert j["n_gpus"] == 3 and j["ranks_seen"] == 3 and j["scaling"] == "strong" and "NOT a measurement" in j["rehearsal"]
    assert j["one_mesh"]["comm_selftest_mismatches"] == 0
    assert j["config"]["one_mesh_form"] == "zslab_indexed_gather" and (j["triangles"], j["vertices"]) == (530536, 265262)
    for form in ("zslab_indexed_gather", "xslab_distributed_weld"):
        assert j["one_mesh"][form]["512^3"]["triangles"] == 530536 and j["one_mesh"][form]["256^3"]["route"].startswith("planned")
    assert len(j["one_mesh"]["slab_bounds"]["z"]) == 4 and j["replicas"]["ms_per_step"] > 0
    assert j["dense_p1_sharded"]["512^3"]["sphere"]["planes_of_rank0"] == [0, 171]
