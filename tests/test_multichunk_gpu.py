"""GPU: the CHUNKED drivers against the reference's own multi-chunk run (tests/golden/e2e_multichunk.npz, made by
tools/gen_golden_multichunk.py from /root/reference: 16 901 surface points = four 5000-point chunks with a ragged last one,
and refine on exactly 10 000 points).

SURVEY.md §8 row a14: refine's smoothing loop is Gauss-Seidel over the chunks -- `Neighbor_ori = ori[index]`
(/root/reference/PMVO.py:612) reads what earlier chunks wrote back (:640).  Every form of the loop the product has is pinned
here to the REFERENCE's files, not to another form of itself:
  * one rank, device-resident pass (round 6, the default: medoid -> replacement chain on the main stream, losses in groups of
    chunks on a side stream, threshold / shell points / voxel fit without leaving the device);
  * one rank, host-driven split chain (MH_REFINE_DEVICE=0: the default of rounds 4-5);
  * one rank, four launches per chunk (MH_REFINE_CHAIN=0: the form the sharded path runs);
  * 2 and 3 ranks (gloo ranks sharing the test GPU), every rank owning a slice of every chunk, one in-place all_gather per chunk.
optimize over four chunks on three rotating streams is pinned the same way (row a10)."""
import ast
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
HELPER = os.path.join(ROOT, "tests", "golden_drivers.py")


def golden():
    z = np.load(os.path.join(GOLDEN, "e2e_multichunk.npz"), allow_pickle=False)
    return z, ast.literal_eval(str(z["meta"]))


def same_rows(a, b):
    s = (a == b) | (np.isnan(a) & np.isnan(b))
    return s if s.ndim == 1 else s.all(axis=1)


def check_refine_files(out, z, prefix, n):
    import scipy.io

    mat = "" if prefix == "ref_" else prefix           # (the main run's voxel arrays carry no prefix in the fixture)

    r = {k: np.load(os.path.join(out, "refine", k + ".npy")) for k in
         ("select_p", "select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori")}
    assert len(r["select_o"]) == n
    om = same_rows(r["select_o"], z[prefix + "select_o"])
    assert om.all(), ("orientations", float(om.mean()), np.flatnonzero(~om)[:5])
    lm = same_rows(r["min_loss"], z[prefix + "min_loss"])
    assert lm.all(), ("losses", float(lm.mean()), np.flatnonzero(~lm)[:5])   # every row: the chunks' trailing points included
    assert np.array_equal(r["filter_unvisible"], z[prefix + "filter_unvisible"])
    fm = same_rows(r["filter_unvisible_ori"], z[prefix + "filter_unvisible_ori"])
    assert fm.all(), float(fm.mean())                       # every shell point's orientation (rows a15)
    Occ3 = scipy.io.loadmat(os.path.join(out, "refine", "Occ3D.mat"))["Occ"]
    Ori3 = scipy.io.loadmat(os.path.join(out, "refine", "Ori3D.mat"))["Ori"]
    nz = np.argwhere(Occ3 != 0).astype(np.int32)
    ref_nz = z[mat + "mat_occ_nz"]
    a, b = set(map(tuple, nz.tolist())), set(map(tuple, ref_nz.tolist()))
    assert a == b, (len(a), len(b), len(a ^ b))             # the same occupied voxels
    Z = Occ3.shape[2]
    got = np.stack([Ori3[ref_nz[:, 0], ref_nz[:, 1], c * Z + ref_nz[:, 2]] for c in range(3)], 1)
    vm = np.all(got == z[mat + "mat_ori_at_nz"], axis=1)
    assert vm.all(), float(vm.mean())                       # every voxel's orientation, bit for bit


def run_helper(out, what, ranks=1, env_extra=None, port=29600):
    env = dict(os.environ, PYTHONPATH=ROOT, **(env_extra or {}))
    if ranks == 1:
        cmd = [sys.executable, HELPER, "--out", str(out), "--what", what]
    else:
        env.update(MH_DIST_BACKEND="gloo", MH_DEVICE_OVERRIDE="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), HELPER, "--out", str(out), "--what", what]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


FORM_ENV = {"chain": {"MH_REFINE_CHAIN": "1", "MH_REFINE_DEVICE": "1"},
            "host_chain": {"MH_REFINE_CHAIN": "1", "MH_REFINE_DEVICE": "0"},
            "four_launch": {"MH_REFINE_CHAIN": "0"}}


@pytest.mark.parametrize("form", ["chain", "host_chain", "four_launch"])
def test_refine_four_chunks_equals_the_reference(tmp_path, form):
    z, meta = golden()
    run_helper(tmp_path, "refine,refine_exact", env_extra=FORM_ENV[form])
    check_refine_files(os.path.join(tmp_path, "run"), z, "ref_", 16901)
    # N = 10 000: `step = N // 5000 + 1` (PMVO.py:603) makes a third, empty chunk; the reference runs through it and so do we
    assert str(z["exact_raised"]) == ""
    check_refine_files(os.path.join(tmp_path, "exact"), z, "exact_", meta["exact"])


@pytest.mark.parametrize("ranks", [2, 3])
def test_refine_four_chunks_sharded_over_ranks_equals_the_reference(tmp_path, ranks):
    """every rank owns ceil(n/ranks) rows of every chunk (3 ranks: slices that do not divide 5000 or 1901)"""
    z, meta = golden()
    run_helper(tmp_path, "refine,refine_exact", ranks=ranks, env_extra={"MH_REFINE_SHARD": "1"}, port=29600 + ranks)
    check_refine_files(os.path.join(tmp_path, "run"), z, "ref_", 16901)
    check_refine_files(os.path.join(tmp_path, "exact"), z, "exact_", meta["exact"])


@pytest.mark.parametrize("ranks", [1, 2])
def test_optimize_four_chunks_equals_the_reference(tmp_path, ranks):
    """optimize (PMVO.py:565-595) over four chunks rotating over three HIP streams (one rank) / dealt to two ranks: ALL
    16 901 rows of the reference's own four-chunk files, bit for bit (the kernels follow the (rank, base view) group sizes of
    each chunk as MKL's sgemm does in the reference; with the mid forms for every point 135 rows differ)."""
    from conftest import rows_equal

    z, meta = golden()
    run_helper(tmp_path, "optimize,optimize_exact", ranks=ranks, port=29610 + ranks)
    got = {k: np.load(os.path.join(tmp_path, "run", "optimize", k + ".npy")) for k in
           ("select_p", "select_o", "min_loss", "high_conf_index")}
    assert got["select_p"].dtype == np.float32 and got["high_conf_index"].dtype == np.bool_
    assert np.array_equal(got["select_p"], z["opt_select_p"])
    eq = rows_equal((got["select_o"], got["min_loss"], got["high_conf_index"]),
                    (z["opt_select_o"], z["opt_min_loss"], z["opt_high_conf_index"]))
    assert len(eq) == 16901 and eq.all(), (int((~eq).sum()), np.flatnonzero(~eq)[:10])
    # exactly 10 000 points: the reference walks a third, empty chunk and writes the prefix of the four-chunk run
    # (recorded: exact_opt_raised == '', exact_opt_equal_prefix); so do we
    assert str(z["exact_opt_raised"]) == "" and bool(z["exact_opt_equal_prefix"])
    n = meta["exact"]
    ex = {k: np.load(os.path.join(tmp_path, "exact", "optimize", k + ".npy")) for k in
          ("select_p", "select_o", "min_loss", "high_conf_index")}
    for k in ex:
        assert len(ex[k]) == n and np.array_equal(ex[k], got[k][:n], equal_nan=(k != "high_conf_index")), k


@pytest.mark.parametrize("form,ranks", [("chain", 1), ("host_chain", 1), ("four_launch", 1), ("sharded", 2)])
def test_refine_head_filtered_and_nan_rows_equal_the_reference(tmp_path, form, ranks):
    """tests/golden/e2e_headfilter.npz: 6000 points (two chunks), a third head-filtered (loss -1 -> 0.5, PMVO.py:91-92,639),
    40 rows of NaN orientation / loss among the inputs -- the reference's files, in every form of the loop."""
    z = np.load(os.path.join(GOLDEN, "e2e_headfilter.npz"), allow_pickle=False)
    env = dict(FORM_ENV.get(form, {"MH_REFINE_CHAIN": "1"}))
    if ranks > 1:
        env["MH_REFINE_SHARD"] = "1"
    run_helper(tmp_path, "refine_headfilter", ranks=ranks, env_extra=env, port=29640)
    out = os.path.join(tmp_path, "headfilter", "refine")
    got_o, got_l = np.load(os.path.join(out, "select_o.npy")), np.load(os.path.join(out, "min_loss.npy"))
    ref_o, ref_l = z["ref_select_o"], z["ref_min_loss"]
    assert same_rows(got_o, ref_o).all()
    lm = same_rows(got_l, ref_l)
    assert lm.all(), np.flatnonzero(~lm)[:10]
    assert np.array_equal(got_l == 0.5, ref_l == 0.5) and (ref_l == 0.5).sum() == 2000
    assert np.array_equal(np.isnan(got_l), np.isnan(ref_l)) and np.isnan(ref_l).sum() > 0
    assert np.array_equal(np.load(os.path.join(out, "filter_unvisible.npy")), z["ref_filter_unvisible"])
    assert np.array_equal(np.load(os.path.join(out, "filter_unvisible_ori.npy")), z["ref_filter_unvisible_ori"], equal_nan=True)


@pytest.mark.parametrize("threads", [1, 2, 4])
def test_forward_at_other_reference_thread_counts(threads):
    """tests/golden/pmvo_threads.npz (tools/gen_golden_threads.py): the reference's forward() on the first chunk of the
    four-chunk run under torch.set_num_threads(1 / 2 / 4).  The option reproject_fma_min_cols, set to the value the fixture
    recorded for that thread count (tools/probe_mkl_forms.py --emit-options on the reference's host), makes the kernels
    round as that host's MKL does: every row of (orientation, loss, flag), 70 / 494 / 1 780 of which differ from the
    8-thread answer the default reproduces."""
    import torch
    from conftest import golden_records, golden_scene, rows_equal
    from monohair_amd.pmvo import PMVO

    z, meta = golden()
    t = np.load(os.path.join(GOLDEN, "pmvo_threads.npz"), allow_pickle=False)
    info = ast.literal_eval(str(t["meta"]))["by_threads"][threads]
    dev = torch.device("cuda", 0)
    scene = golden_scene(meta)
    pm = PMVO.from_planes(golden_records(z), scene["depth"].to(dev), scene["ori"].to(dev), scene["conf"].to(dev),
                          scene["mask"].to(dev), device=dev, patch_size=meta["patch"], visible_threshold=meta["vis_thr"],
                          conf_threshold=meta["thr"])
    pts = z["opt_select_p"][:5000]
    ref = (t["t%d_ori" % threads], t["t%d_loss" % threads], t["t%d_hc" % threads])
    eight = (z["opt_select_o"][:5000], z["opt_min_loss"][:5000], z["opt_high_conf_index"][:5000])

    def fwd():
        _, o, l, h = pm.forward(pts)
        return o.cpu().numpy(), l.cpu().numpy(), h.cpu().numpy()

    assert rows_equal(fwd(), eight).all()
    pm.set_option("reproject_fma_min_cols", info["reproject_fma_min_cols"])
    eq = rows_equal(fwd(), ref)
    assert eq.all(), (threads, int((~eq).sum()))
    assert int((~rows_equal(ref, eight)).sum()) == info["rows_differing_from_8_threads"]


@pytest.mark.parametrize("prefetch", ["1", "0"])
def test_whole_pass_through_the_three_drivers_equals_the_reference(tmp_path, prefetch):
    """filter_negative_points -> optimize -> refine on the reference's candidates, the arrays handed from one driver to the next
    (PMVO.py:847-873).  With the prefetch on, refine adopts the neighbour table, the head votes and the .mat writer that
    optimize prepared while it iterated (asserted: that path ran); off, it prepares them itself.  Either way: the reference's
    masks, its four optimize files on all 16 901 rows, its five refine files and every voxel."""
    import json

    from conftest import rows_equal

    z, meta = golden()
    run_helper(tmp_path, "pass", env_extra={"MH_REFINE_PREFETCH": prefetch, "MH_MAT_EARLY": prefetch})
    out = os.path.join(tmp_path, "pass")
    info = json.load(open(os.path.join(out, "last_refine.json")))
    # (on this scene some shell queries need another cell size: the device shell stage retries them -- GridKNN's own retries on
    # the rows the first cell size left unfinished -- and stays on the device; the no-retry form runs, asserted, in
    # tests/test_full_size_refine_gpu.py)
    assert info["device_pass"] and info["prefetch_adopted"] == (prefetch == "1"), info
    assert info["shell_stage"].startswith("device"), info
    assert np.array_equal(np.load(os.path.join(out, "optimize", "surface_index.npy")), z["surface_index"])
    assert np.array_equal(np.load(os.path.join(out, "optimize", "filter_index.npy")), z["filter_index"])
    got = {k: np.load(os.path.join(out, "optimize", k + ".npy")) for k in ("select_p", "select_o", "min_loss", "high_conf_index")}
    assert np.array_equal(got["select_p"], z["opt_select_p"])
    eq = rows_equal((got["select_o"], got["min_loss"], got["high_conf_index"]),
                    (z["opt_select_o"], z["opt_min_loss"], z["opt_high_conf_index"]))
    assert len(eq) == 16901 and eq.all()
    check_refine_files(out, z, "ref_", 16901)
    assert not [f for f in os.listdir(os.path.join(out, "refine")) if f.endswith(".writing")]
