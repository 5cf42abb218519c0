"""GPU: the CHUNKED drivers at BASELINE.json's headline size -- bench.py's own scene (60 views @ 1920x1080, continuous maps,
the 256^3 candidate grid: ~290 k surface points = 58 chunks of 5000) -- against the CPU oracle.

  * optimize(): all 58 chunks rotating over three HIP streams, results written straight into their slices; three whole
    chunks (first, a middle one, the ragged last) and a random sample of 2000 rows are compared with oracle.forward.
  * refine(): the Gauss-Seidel smoothing loop over 58 chunks (/root/reference/PMVO.py:602-643) against
    oracle.refine_loop -- which tests/test_oracle_more.py pins to the reference's own four-chunk run -- on EVERY point, bit for
    bit: orientations and losses (HIP and oracle add in the same order; no N-mod-64 exception here)."""
import os
import types

import numpy as np
import pytest
import torch
from scipy.spatial import KDTree

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def headline(tmp_path_factory):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO, filter_negative_points, optimize

    V, H, W, patch, thr = 60, 1920, 1080, 7, 0.15
    scene = synth.make_scene(V, H, W, device=DEV, seed=0)                 # continuous maps: what bench.py times
    cams = cameras_from_list(scene["cams"])
    rec = camera_records(cams)
    pm = PMVO.from_planes(rec, scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=DEV,
                          patch_size=patch, visible_threshold=1, conf_threshold=thr, camera=cams)
    rngb = np.random.default_rng(123)
    bust = rngb.normal(size=(2000, 3))
    bust = bust / np.linalg.norm(bust, axis=1, keepdims=True) * 0.09
    scalp = bust[bust[:, 1] > 0.03] * (0.1 / 0.09)
    pm.set_head(KDTree(data=bust), KDTree(data=scalp), np.max(scalp, axis=0))
    root = tmp_path_factory.mktemp("headline")
    args = types.SimpleNamespace(device=DEV, output_path=str(root), save_root=str(root / "optimize"),
                                 save_path=str(root / "refine"), PMVO=types.SimpleNamespace(visible_threshold=1.0),
                                 data=types.SimpleNamespace(root=str(root)))
    os.makedirs(args.save_path, exist_ok=True)
    cand = synth.candidate_points(res=256, seed=0)
    surface_index, surface_points, filter_index = filter_negative_points(cand, pm, args)
    assert 250000 < len(surface_points) < 350000 and len(surface_points) % 5000 != 0
    optimize(surface_points, pm, args)
    opt = {k: np.load(os.path.join(args.save_root, k + ".npy")) for k in ("select_p", "select_o", "min_loss",
                                                                           "high_conf_index")}
    views = oracle.Views(rec, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(), scene["conf"].cpu().numpy(),
                         scene["mask"].cpu().numpy())
    shell = cand[:len(filter_index)][filter_index]
    return dict(pm=pm, args=args, opt=opt, views=views, patch=patch, thr=thr, scalp=scalp, shell=shell)


def eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_optimize_58_chunks_vs_oracle(headline):
    from monohair_amd.pmvo import depth_offsets

    h = headline
    opt, views = h["opt"], h["views"]
    N = len(opt["select_p"])
    nchunk = N // 5000 + 1
    assert nchunk >= 58 and opt["select_p"].dtype == np.float32
    offs = depth_offsets(90)
    for c in (0, nchunk // 2, nchunk - 1):                    # whole chunks, the ragged last one included
        lo, hi = c * 5000, min((c + 1) * 5000, N)
        _, o_ori, o_loss, o_hc = oracle.forward(views, opt["select_p"][lo:hi], h["patch"], h["thr"], offs)
        assert eq(opt["min_loss"][lo:hi], o_loss) and eq(opt["select_o"][lo:hi], o_ori), c
        assert eq(opt["high_conf_index"][lo:hi], o_hc), c
    # (a point's answer depends on its chunk -- group sizes per base view, trailing columns of the sums -- so chunks are
    # compared with chunks; two more, picked at random)
    for c in np.random.default_rng(5).choice(nchunk - 1, 2, replace=False):
        lo, hi = int(c) * 5000, (int(c) + 1) * 5000
        _, o_ori, o_loss, o_hc = oracle.forward(views, opt["select_p"][lo:hi], h["patch"], h["thr"], offs)
        assert eq(opt["min_loss"][lo:hi], o_loss) and eq(opt["select_o"][lo:hi], o_ori) and eq(opt["high_conf_index"][lo:hi], o_hc)
    assert np.isfinite(opt["min_loss"]).mean() > 0.9


@pytest.mark.parametrize("form", ["chain", "four_launch"])
def test_refine_loop_58_chunks_vs_oracle(headline, form, monkeypatch):
    from monohair_amd.pmvo import refine

    h = headline
    opt = h["opt"]
    monkeypatch.setenv("MH_REFINE_CHAIN", "1" if form == "chain" else "0")
    pts, ori, loss = opt["select_p"].copy(), opt["select_o"].copy(), opt["min_loss"].copy()
    refine(pts, ori, loss, h["pm"], h["shell"][:20000].copy(), h["args"], infer_inner=False, threshold=0.001,
           genrate_ori_only=False, return_dense=False)
    got_o = np.load(os.path.join(h["args"].output_path, "refine", "select_o.npy"))
    got_l = np.load(os.path.join(h["args"].output_path, "refine", "min_loss.npy"))
    want_o, want_l = opt["select_o"].copy(), opt["min_loss"].copy()
    trace = []
    scalp = h["scalp"]
    oracle.refine_loop(h["views"], opt["select_p"], want_o, want_l, h["patch"], h["thr"], 1.0, KDTree(data=scalp),
                       np.max(scalp, axis=0), trace=trace)
    assert len(trace) >= 58 and sum(r for _, _, r in trace) > 100            # the loop does replace orientations
    om = np.all((got_o == want_o) | (np.isnan(got_o) & np.isnan(want_o)), axis=1)
    lm = (got_l == want_l) | (np.isnan(got_l) & np.isnan(want_l))
    assert om.all() and lm.all(), (float(om.mean()), float(lm.mean()), np.flatnonzero(~om)[:5], np.flatnonzero(~lm)[:5])
    print("refine loop at the headline size (%s): %d points, %d chunks, %d replaced, %d head-filtered (loss 0.5)"
          % (form, len(want_l), len(trace), sum(r for _, _, r in trace), int((want_l == 0.5).sum())))
    # Gauss-Seidel, not Jacobi: the result differs from medoids taken from the INPUT orientations
    assert not eq(got_o, opt["select_o"])


def test_device_resident_refine_writes_the_files_of_the_host_driven_form(headline, monkeypatch, tmp_path):
    """Round 6: refine()'s default on one rank keeps threshold / kept rows / shell neighbours / concatenation / voxel fit on the
    device (pmvo.py::_refine_device).  At the headline size, with a threshold that keeps most points (0.025, bench.py's), the
    whole of it runs there -- asserted -- and every file it writes (the three smoothed arrays, the kept shell points and their
    orientations, Ori3D.mat / Occ3D.mat) is byte for byte the file of the host-driven form of rounds 4-5 (MH_REFINE_DEVICE=0),
    whose shell stage, voxel fit and files are pinned to the reference's (tests/test_multichunk_gpu.py)."""
    import hashlib

    from monohair_amd.pmvo import refine

    h = headline
    opt = h["opt"]
    digests, info = {}, {}
    for form in ("device", "host"):
        monkeypatch.setenv("MH_REFINE_DEVICE", "1" if form == "device" else "0")
        root = tmp_path / form
        args = types.SimpleNamespace(device=DEV, output_path=str(root), save_root=str(root / "optimize"),
                                     save_path=str(root / "refine"), PMVO=types.SimpleNamespace(visible_threshold=1.0),
                                     data=types.SimpleNamespace(root=str(root)))
        os.makedirs(args.save_path, exist_ok=True)
        refine(opt["select_p"].copy(), opt["select_o"].copy(), opt["min_loss"].copy(), h["pm"], h["shell"].copy(), args,
               infer_inner=False, threshold=0.025, genrate_ori_only=False, return_dense=False)
        info[form] = dict(h["pm"].last_refine)
        names = sorted(os.listdir(root / "refine"))
        digests[form] = {n: hashlib.sha256(open(root / "refine" / n, "rb").read()).hexdigest() for n in names}
    assert info["device"] == {"device_pass": True, "prefetch_adopted": False, "shell_stage": "device"}, info
    assert not info["host"]["device_pass"]
    assert sorted(digests["device"]) == ["Occ3D.mat", "Ori3D.mat", "filter_unvisible.npy", "filter_unvisible_ori.npy",
                                         "min_loss.npy", "select_o.npy", "select_p.npy"]
    # (the .mat header carries a creation time: compare the payloads)
    for n in digests["device"]:
        a, b = (open(tmp_path / f / "refine" / n, "rb").read() for f in ("device", "host"))
        if n.endswith(".mat"):
            a, b = a[128:], b[128:]
        assert a == b, n
    kept = np.load(tmp_path / "device" / "refine" / "filter_unvisible.npy")
    assert 1000 < len(kept) <= len(h["shell"])
