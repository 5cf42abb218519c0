"""bench.py's contract on the GPU box: one JSON line; the roofline block describes the kernels of the timed loop;
`python bench.py --gpus 2` starts its two ranks itself (here they share the one GPU over gloo -- RCCL refuses two
ranks on one device -- which still runs the whole multi-rank code path: rank discovery, sharded steps, the sharded
full pass)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SMALL = ["--steps", "3", "--warmup", "1", "--views", "24", "--height", "240", "--width", "136", "--volume", "48",
         "--patch", "3", "--no-cpu"]
SMALL_CPU = [x for x in SMALL if x != "--no-cpu"]


def run_bench(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT, **(env or {})), stdin=subprocess.DEVNULL, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_and_rooflines_of_the_timed_kernels():
    d = run_bench(SMALL)
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["steps"] == 3 and d["unit"] == "iterations/s"
    rf = d["roofline"]
    assert rf["kernel"].startswith("mh_search3_kernel") and rf["bound"] == "valu" and rf["peak"] == 157.3
    assert 0 < rf["pair_evals_executed"] <= rf["pair_evals_nominal"]
    assert rf["tap_body"].startswith("key") and d["secondary_8bit_maps"]["roofline"]["tap_body"].startswith("select")
    # frac must be recomputable from the line itself
    assert abs(rf["pair_evals_executed"] * rf["flop_per_pair_eval"] / (rf["launch_ms"] * 1e-3) / 1e12 - rf["achieved"]) < 0.05
    names = [k["kernel"] for k in d["roofline_kernels"]]
    assert any(n.startswith("mh_project_taps") for n in names) and any(n.startswith("mh_project_gather") for n in names)
    for k in d["roofline_kernels"]:
        assert abs(k["algorithmic_bytes_per_launch"] / (k["launch_ms"] * 1e-3) / 1e9 - k["achieved"]) < 1.0
    assert "refine_and_volume_s" in d["secondary_full_pass"]
    assert d["timed_region_s"] > 0 and abs(d["timed_region_s"] / d["steps"] * 1e3 - d["ms_per_step"]) < 1e-3 * d["ms_per_step"] + 1e-4
    g = d["secondary_gabor_sharded"]
    assert g["n_gpus"] == 1 and g["views"] == 24 and g["value"] > 0 and g["codes_agree_on_all_ranks"] is True


def test_line_verifies_its_own_outputs_against_the_oracle():
    """with the CPU leg: the outputs of the LAST timed step are compared with oracle.forward in the same run -- the WHOLE chunk
    whatever part of it the timed CPU sample covers (a point's answer depends on its batch) -- and whatever the line copies
    from profiles/traffic.json says so"""
    d = run_bench([x for x in SMALL if x != "--no-cpu"] + ["--no-secondary", "--cpu-points", "800"])
    pc = d["parity_check"]
    assert pc["bit_exact"] is True and pc["finite_losses"] > 0
    assert pc["points"] == min(d["config"]["points_per_iteration"], d["config"]["surface_points"]) or pc["points"] > 800
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    rf = d["roofline"]
    assert rf["traffic"] is None or "profiles/traffic.json" in rf["traffic_source"]


def test_gpus_2_spawns_two_ranks_itself():
    d = run_bench(["--gpus", "2"] + SMALL_CPU, env={"MH_DIST_BACKEND": "gloo", "MH_DEVICE_OVERRIDE": "0"})
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["backend"] == "gloo"
    assert d["parity_check"]["bit_exact"] is True and d["parity_check"]["points"] > 0 and "cpu_baseline" not in d
    assert d["secondary_full_pass"]["ranks"] == 2 and "total_s" in d["secondary_full_pass"]
    assert len(d["per_rank_iterations_per_s"]) == 2 and min(d["per_rank_iterations_per_s"]) > 0
    g = d["secondary_gabor_sharded"]
    assert g["n_gpus"] == 2 and g["views_per_rank"] == [12, 12] and g["codes_agree_on_all_ranks"] is True
    assert g["all_gather_bytes"] == 24 * 2 * 240 * 136 and g["all_gather_ms"] > 0


def test_gpus_2_volume_exchange_legs_run_through_the_rccl_stand_in():
    """the C-ABI exchange legs of `bench.py --gpus 2` (and the full pass with MH_VOLUME_EXCHANGE=capi) with two ranks on
    the one GPU: the RCCL entry points are bound to tests/fake_rccl.cpp"""
    from conftest import fake_rccl_lib

    d = run_bench(["--gpus", "2"] + SMALL, env={"MH_DIST_BACKEND": "gloo", "MH_DEVICE_OVERRIDE": "0",
                                                  "MH_RCCL_LIB": fake_rccl_lib(), "MH_VOLUME_EXCHANGE": "capi"})
    v = d["secondary_volume_reduce"]
    assert v["default_exchange"] == "capi"
    for leg in ("slab_gather_torch", "slab_gather_c_abi", "dense_reduce_c_abi"):
        assert v.get(leg + "_correct") is True, v
    assert d["secondary_full_pass"]["ranks"] == 2 and "error" not in d["secondary_full_pass"]


def test_gpus_more_than_present_is_refused():
    import torch

    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + SMALL, cwd=ROOT,
                       env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
