"""Pin the remaining oracle restatements (filters, refine loss, consensus, voxel fit, Gabor bank) against golden
vectors generated from the reference.  CPU only."""
import ast

import numpy as np
import pytest
from scipy.spatial import KDTree

import oracle
from conftest import golden_records, golden_scene, load_golden, scene_views

CASES = ["pmvo_small", "pmvo_mid", "pmvo_quant", "pmvo_views300", "pmvo_views300c", "pmvo_patch9", "pmvo_patch4"]


@pytest.fixture(scope="module", params=CASES)
def case(request):
    meta, z = load_golden(request.param)
    scene = golden_scene(meta)
    return meta, z, scene, scene_views(scene, golden_records(z))


def test_filter_votes(case):
    meta, z, scene, views = case
    surf, filt, unv, _ = oracle.filter_votes(views, z["filter_points_in"], meta["patch"], meta["thr"], meta["vis_thr"])
    assert np.array_equal(surf, z["filter_surface_index"])
    assert np.array_equal(filt, z["filter_filter_index"])
    assert np.array_equal(unv, z["unvisible_index"])
    # every branch is exercised
    assert 0 < surf.sum() < len(surf) and filt.sum() > 0 and 0 < unv.sum() < len(unv)


def head_top_index(pts32, scalp):
    """The host part of filter_head_points (PMVO.py:98-107)."""
    d, _ = KDTree(data=scalp).query(pts32, k=1)
    return np.logical_and(d < 0.04, pts32[:, 2] < np.max(scalp, axis=0)[2] - 0.01)


def test_refine_method_loss(case):
    meta, z, scene, views = case
    pts = z["points"]
    loss, _ = oracle.refine_loss(views, pts, z["refine_ori_in"], meta["patch"], meta["thr"])
    _, _, _, head = oracle.filter_votes(views, pts, meta["patch"], meta["thr"], meta["vis_thr"])
    filt = np.logical_and(head, ~head_top_index(pts.astype(np.float32), z["toy_scalp"]))
    loss = loss.copy()
    loss[filt] = -1
    ref = z["refine_loss"]
    # every row, bit for bit -- the trailing N mod 32 points included, whose [V,N,1] sums ATen adds in its row_sum order
    # (oracle/pmvo_oracle.c: row_sum1)
    assert np.array_equal(loss, ref, equal_nan=True)
    assert (ref == -1).sum() > 0 or meta.get("cluster")      # (12 clustered points: the head filter may hit none)


def test_consensus_medoid():
    """The oracle's medoid (mean in ATen's inner-dimension summation order) against the reference on every golden
    group: 100 % -- exact ties, degenerate groups, group sizes on both sides of every branch of ATen's sum (scalar
    path below 8, leftover vectors, tail elements, the second cascade level from 512 members on) and tight clusters."""
    for name in ("consensus", "consensus_more"):
        z = load_npz(name)
        for k in sorted(f[:-3] for f in z.files if f.endswith("_in")):
            out, idx = oracle.medoid_dense(z[k + "_in"])
            ref = z[k + "_out"]
            ok = np.all((out == ref) | (np.isnan(out) & np.isnan(ref)), axis=1)
            assert ok.all(), (name, k, float(ok.mean()))


def test_aten_inner_sum_order_matches_torch():
    """orc_aten_inner_sum == torch.sum over the last dimension of a contiguous float32 tensor, bit for bit (the claim the
    medoid's parity rests on; torch is the very library the reference calls)."""
    import ctypes

    import torch

    L = oracle.lib()
    L.orc_aten_inner_sum.restype = ctypes.c_float
    L.orc_aten_inner_sum.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rng = np.random.default_rng(9)
    for K in list(range(1, 70)) + [100, 127, 128, 129, 255, 256, 300, 511, 512, 513, 1000, 4097, 9000]:
        x = np.ascontiguousarray(rng.random((3, K)).astype(np.float32) - 0.3)
        want = torch.from_numpy(x).sum(dim=-1).numpy()
        got = np.array([L.orc_aten_inner_sum(x[r].ctypes.data_as(ctypes.c_void_p), K) for r in range(3)], np.float32)
        assert np.array_equal(got, want), K


def load_npz(name):
    import os

    from conftest import GOLDEN

    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def test_voxel_fit_and_mat_layout():
    """refine's volume fit (PMVO.py:656-764) from the reference's own refined points."""
    z = load_npz("e2e_small")
    meta = ast.literal_eval(str(z["meta"]))
    keep = np.where(z["ref_min_loss"] < meta["threshold"])[0]
    sel_o = np.concatenate([z["ref_select_o"][keep], z["ref_filter_unvisible_ori"]], 0)
    sel_p = np.concatenate([z["ref_select_p"][keep], z["ref_filter_unvisible"]], 0)
    occ, ori = oracle.voxel_fit(sel_p.copy(), sel_o.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
    ori_l, occ_l = oracle.mat_layout(occ, ori)
    assert tuple(z["mat_ori_shape"]) == ori_l.shape and tuple(z["mat_occ_shape"]) == occ_l.shape
    nz = np.argwhere(occ_l != 0).astype(np.int32)
    assert np.array_equal(nz, z["mat_occ_nz"])
    Z = occ_l.shape[2]
    got = np.stack([ori_l[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    same = np.all(got == z["mat_ori_at_nz"], axis=1)
    assert same.all(), same.mean()            # per-voxel medoid in ATen's summation order: every voxel, bit for bit
    assert int(z["mat_ori_nnz"][0]) == np.count_nonzero(ori_l)


def test_infer_inner_merge_and_mat_files(tmp_path):
    """PMVO.py:733-751 -- the DeepMVSHair points no view sees overwrite the fitted volume ("later rows win", also among
    themselves), y > 0 orientations flipped first, coarse.npy / coarse_ori.npy -- against the reference's own second
    pass (tests/golden/e2e_inner.npz: exterior voxels of e2e_small + a synthetic ours/raw.npy).  Host logic only:
    merge_inner_points + the sparse MAT-v5 writer, read back with scipy."""
    import scipy.io

    from monohair_amd import pmvo_utils as U

    ext, inn = load_npz("e2e_small"), load_npz("e2e_inner")
    nz = ext["mat_occ_nz"].astype(np.int64)                    # Occ[Y,X,Z] indices of the exterior pass
    vox = nz[:, [1, 0, 2]]
    vori = ext["mat_ori_at_nz"]
    vox, vori, coarse, coarse_ori = U.merge_inner_points(vox, vori, inn["raw"], inn["unvisible_index"])
    assert coarse.dtype == np.float32 and np.array_equal(coarse, inn["coarse"])
    assert coarse_ori.dtype == np.float32 and np.array_equal(coarse_ori, inn["coarse_ori"])
    assert 0 < inn["unvisible_index"].sum() < len(inn["raw"]) and (inn["coarse_ori"][:, 1] <= 0).all()
    U.save_ori_occ_mat_sparse(str(tmp_path), U.GRID_RESOLUTION, vox, vori)
    Ori3 = scipy.io.loadmat(str(tmp_path / "Ori3D.mat"))["Ori"]
    Occ3 = scipy.io.loadmat(str(tmp_path / "Occ3D.mat"))["Occ"]
    assert Ori3.shape == tuple(inn["mat_ori_shape"]) and Occ3.shape == tuple(inn["mat_occ_shape"])
    got_nz = np.argwhere(Occ3 != 0).astype(np.int32)
    assert np.array_equal(got_nz, inn["mat_occ_nz"])
    Z = Occ3.shape[2]
    got = np.stack([Ori3[got_nz[:, 0], got_nz[:, 1], c * Z + got_nz[:, 2]] for c in range(3)], 1)
    assert np.array_equal(got, inn["mat_ori_at_nz"])
    assert int(inn["mat_ori_nnz"][0]) == np.count_nonzero(Ori3)
    # the fixture really exercises the overwrite rules: raw rows colliding with fitted voxels and with each other
    x, y, z = U.p2v(inn["coarse"].copy(), U.VOXEL_MIN, U.VOXEL_SIZE, U.GRID_RESOLUTION)
    keys = (x.astype(np.int64) * 256 + y) * 192 + z
    assert len(np.unique(keys)) < len(keys)
    assert len(set(map(tuple, nz[:, [1, 0, 2]].tolist())) & set(zip(x.tolist(), y.tolist(), z.tolist()))) > 0


def conf_code(conf):
    """the 8-bit code the reference writes for a confidence (preprocess_capture_data/GaborFilter.py:210: torchvision's
    save_image = mul(255).add_(0.5).clamp_(0, 255) in float32, then the truncating cast to uint8)"""
    c = conf.astype(np.float32) * np.float32(255.0) + np.float32(0.5)
    return np.clip(c, 0, 255).astype(np.uint8)


def test_gabor_bank_vs_reference():
    z = load_npz("gabor")
    bank = z["bank"]
    # what PMVO reads is the 8-bit FILE of the confidence: the <= 1-ulp differences below must not cross a code boundary
    for name in ("stripes0", "stripes30", "stripes90", "stripes135", "noise", "mixed"):
        _, conf, _ = oracle.gabor_bank(bank, z[name + "_img"])
        assert np.array_equal(conf_code(conf), conf_code(z[name + "_conf"])), name      # every pixel's code
    for name, want in (("stripes0", 0), ("stripes30", 30), ("stripes90", 90), ("stripes135", 135)):
        orient, conf, var = oracle.gabor_bank(bank, z[name + "_img"])
        ref_idx = np.rint(z[name + "_best"] * 180.0 / np.pi).astype(np.int32)
        inner = (slice(12, -12), slice(12, -12))
        # known answer: stripes with oscillation axis (row,col)=(cos t, sin t) -> index t (period-4 stripes
        # alias on a few pixels at oblique angles, in the reference too)
        assert (ref_idx[inner] == want).mean() >= 0.9
        assert (orient[inner] == want).mean() >= 0.9
        assert np.array_equal(orient, ref_idx), name                 # every pixel's orientation index
        assert np.allclose(conf, z[name + "_conf"], rtol=0, atol=1.2e-7), name      # one float32 ulp at most ...
        assert (conf == z[name + "_conf"]).mean() >= 0.998, name                     # ... on < 0.2 % of the pixels
    for name in ("noise", "mixed"):
        orient, conf, var = oracle.gabor_bank(bank, z[name + "_img"])
        ref_idx = np.rint(z[name + "_best"] * 180.0 / np.pi).astype(np.int32)
        assert np.array_equal(orient, ref_idx), name
        assert np.allclose(conf, z[name + "_conf"], rtol=0, atol=1.2e-7), name
        assert (conf == z[name + "_conf"]).mean() >= 0.998, name


def test_gabor_confidence_is_exact_up_to_the_root_of_mkl():
    """Where the <= 1 ulp of the Gabor confidence comes from: `variance ** (1 / 2)` (GaborFilter.py:77) is handed to MKL's
    vector math library by ATen (vsSqrt, high-accuracy mode: below 1 ulp, not correctly rounded), everything else -- the 289-tap
    correlations, the argmax, the cascade sum under the root, the two divisions -- is restated exactly.  Asserted: torch's own
    root applied to the ORACLE's sums gives the reference's confidence on EVERY pixel of every golden image.  (The root is
    host code of a closed-source library: the comparison runs where torch.sqrt reproduces the values recorded with the goldens,
    tests/golden/mkl_forms.npz, and is skipped elsewhere.)"""
    import torch
    from conftest import GOLDEN

    rec = np.load(__import__("os").path.join(GOLDEN, "mkl_forms.npz"))
    if not np.array_equal(torch.sqrt(torch.from_numpy(rec["sqrt_x"])).numpy(), rec["sqrt_out"]):
        pytest.skip("this host's MKL rounds sqrt differently from the one the goldens were generated with")
    assert (rec["sqrt_out"] != np.sqrt(rec["sqrt_x"])).sum() > 0          # ... and that root is not the IEEE one
    z = load_npz("gabor")
    for name in ("stripes0", "stripes30", "stripes90", "stripes135", "noise", "mixed"):
        _, conf, var, vsum = oracle.gabor_bank(z["bank"], z[name + "_img"], want_sum=True)
        assert np.array_equal(np.sqrt(vsum), var)
        v = torch.from_numpy(vsum)[None, None] ** (1 / 2)
        ref_like = ((v / torch.max(v) - 0) / (0.2 - 0)).clamp(0, 1)[0, 0].numpy()
        assert np.array_equal(ref_like, z[name + "_conf"]), name


# ------------------------------------------------------------------------------------------------------------------
# The CHUNKED drivers (tests/golden/e2e_multichunk.npz, tools/gen_golden_multichunk.py): the reference's own optimize +
# refine over 16 901 surface points = four 5000-point chunks, ragged last, and refine on exactly 10 000 points.
# ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def multichunk():
    z = load_npz("e2e_multichunk")
    meta = ast.literal_eval(str(z["meta"]))
    scene = golden_scene(meta)
    return meta, z, scene_views(scene, golden_records(z))


def loss_rows_equal(got, ref):
    """Losses bit for bit on EVERY row (NaN == NaN) -- the trailing (chunk length mod 32) points of each chunk included, whose
    [V,N,1] sums ATen adds in its row_sum order (oracle/pmvo_oracle.c: row_sum1)."""
    same = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), (float(same.mean()), np.flatnonzero(~same)[:10])
    return same


def test_refine_loop_multichunk_vs_reference(multichunk):
    """oracle.refine_loop == the reference's smoothing loop over FOUR chunks (PMVO.py:602-643): every orientation bit for
    bit.  The loop is Gauss-Seidel -- chunk k+1's medoids read the orientations chunk k replaced (:612,:640) -- and this run
    shows it: a Jacobi reading (all medoids from the input orientations) gives different bits on later chunks."""
    meta, z, views = multichunk
    pts = z["opt_select_p"]
    assert len(pts) == 16901 and pts.dtype == np.float32
    scalp = z["toy_scalp"]
    ori, loss = z["opt_select_o"].copy(), z["opt_min_loss"].copy()
    trace = []
    oracle.refine_loop(views, pts, ori, loss, meta["patch"], meta["thr"], meta["vis_thr"], KDTree(data=scalp),
                       np.max(scalp, axis=0), trace=trace)
    assert [(lo, hi) for lo, hi, _ in trace] == [(0, 5000), (5000, 10000), (10000, 15000), (15000, 16901)]
    assert all(r > 0 for _, _, r in trace)                          # every chunk replaces some orientations
    ref_o, ref_l = z["ref_select_o"], z["ref_min_loss"]
    om = np.all((ori == ref_o) | (np.isnan(ori) & np.isnan(ref_o)), axis=1)
    assert om.all(), float(om.mean())
    loss_rows_equal(loss, ref_l)
    # (no surface point of this scene is head-filtered; the -1 -> 0.5 branch is pinned by test_refine_method_loss)

    # the Jacobi reading is NOT what the reference does: it must differ on chunks 1.. (and agree on chunk 0)
    tree = KDTree(data=pts)
    _, index = tree.query(pts, 100, workers=-1)
    centre_j, _ = oracle.medoid_dense(z["opt_select_o"][index])
    jac = z["opt_select_o"].copy()
    oracle.replace_dissimilar(centre_j, jac, 0.95)
    jm = np.all((jac == ref_o) | (np.isnan(jac) & np.isnan(ref_o)), axis=1)
    assert jm[:5000].all() and not jm[5000:].all(), (jm[:5000].mean(), jm[5000:].mean())


def test_refine_loop_exact_multiple_of_chunk_vs_reference(multichunk):
    """N = 10 000: `step = N // 5000 + 1` (PMVO.py:603) gives a third chunk of zero points.  The reference runs through it
    (recorded: exact_raised == ''; an empty KDTree query, an empty medoid and an empty write-back), so skipping it is the
    same result -- asserted against the reference's files for that run."""
    meta, z, views = multichunk
    assert str(z["exact_raised"]) == "" and str(z["exact_opt_raised"]) == "" and bool(z["exact_opt_equal_prefix"])
    n = meta["exact"]
    pts = z["opt_select_p"][:n]
    scalp = z["toy_scalp"]
    ori, loss = z["opt_select_o"][:n].copy(), z["opt_min_loss"][:n].copy()
    trace = []
    oracle.refine_loop(views, pts, ori, loss, meta["patch"], meta["thr"], meta["vis_thr"], KDTree(data=scalp),
                       np.max(scalp, axis=0), trace=trace)
    assert [(lo, hi) for lo, hi, _ in trace] == [(0, 5000), (5000, 10000)]
    ref_o, ref_l = z["exact_select_o"], z["exact_min_loss"]
    assert np.all((ori == ref_o) | (np.isnan(ori) & np.isnan(ref_o)))
    loss_rows_equal(loss, ref_l)
    # not the prefix of the 16 901-point run: the neighbourhoods near the cut differ
    assert not np.array_equal(ref_o, z["ref_select_o"][:n], equal_nan=True)


def test_optimize_multichunk_vs_reference(multichunk):
    """oracle.forward chunk by chunk (5000 points, PMVO.py:565-579) against the reference's optimize over four chunks: ALL
    16 901 rows of select_o / min_loss / high_conf_index, bit for bit.  A point's answer depends on its chunk: at this size
    18-45 % of a chunk's (rank, point) items share their base view with more than 316 other points, where MKL's sgemm in
    Camera.reprojection (Utils/Camera_utils.py:103) rounds as an fma chain; the oracle follows the group sizes.  With the
    batch-independent option ("mid" forms, rounds 1-4) 135 rows differ, all of them rows on which the reference disagrees
    with itself under recomposition (conftest.check_rows_against_recomposed)."""
    from conftest import GOLDEN, check_rows_against_recomposed, rows_equal

    meta, z, views = multichunk
    pts = z["opt_select_p"]
    offs = np.load(__import__("os").path.join(GOLDEN, "depth_offsets.npy"))
    ref = (z["opt_select_o"], z["opt_min_loss"], z["opt_high_conf_index"])

    def run():
        parts = [oracle.forward(views, pts[a:a + 5000], meta["patch"], meta["thr"], offs)[1:] for a in range(0, len(pts), 5000)]
        return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))

    got = run()
    assert len(got[1]) == 16901 and rows_equal(got, ref).all()
    # the largest (rank, base view) group of this run is well past the 316-point switch
    bidx, _ = oracle.topk_views(*[oracle.visible_and_ori(views, pts[:5000], meta["patch"])[k] for k in ("visible", "Conf")])
    assert max(oracle.group_sizes(bidx[r], views.V).max() for r in range(0, 20, 2)) > 316
    prev = oracle.set_reproject_rule("mid"), oracle.set_sum_block(0)
    try:
        mid = run()
    finally:
        oracle.set_reproject_rule(*prev[0])
        oracle.set_sum_block(prev[1])
    st = check_rows_against_recomposed("e2e_multichunk optimize (mid forms)", mid, ref,
                                       [(z["optrec_select_o"], z["optrec_min_loss"], z["optrec_high_conf_index"])])
    assert st["differ_from_original_batch"] == 135 and st["reference_self_disagreement_over_1e4"] == 2


@pytest.mark.parametrize("threads", [1, 2, 4])
def test_forward_at_other_reference_thread_counts(multichunk, threads):
    """tests/golden/pmvo_threads.npz (tools/gen_golden_threads.py): the reference's forward() on the first 5000-point chunk of
    the four-chunk run under torch.set_num_threads(1 / 2 / 4).  The column count from which MKL's sgemm takes the fma-chain
    form in Camera.reprojection (Utils/Camera_utils.py:81-106) is the switch to its THREADED kernel and moves with the thread
    count of the reference's host (1: never, 2: 21 334, 4: 14 223, 8: 28 445 columns); the fixture stores the value probed at
    each count next to the reference's outputs.  With the rule set from the fixture the oracle equals the reference on every
    row -- 70 / 494 / 1 780 of which differ from the 8-thread files -- and with the 8-thread default it does not."""
    from conftest import GOLDEN, rows_equal

    meta, z, views = multichunk
    t = load_npz("pmvo_threads")
    tm = ast.literal_eval(str(t["meta"]))
    assert tm["case"] == meta and tm["rows"] == 5000
    info = tm["by_threads"][threads]
    pts = z["opt_select_p"][:5000]
    offs = np.load(__import__("os").path.join(GOLDEN, "depth_offsets.npy"))
    ref = (t["t%d_ori" % threads], t["t%d_loss" % threads], t["t%d_hc" % threads])
    eight = (z["opt_select_o"][:5000], z["opt_min_loss"][:5000], z["opt_high_conf_index"][:5000])
    assert int((~rows_equal(ref, eight)).sum()) == info["rows_differing_from_8_threads"] > 0
    assert rows_equal(oracle.forward(views, pts, meta["patch"], meta["thr"], offs)[1:], eight).all()      # the default: 8 threads
    prev = oracle.set_reproject_rule("group", info["reproject_fma_min_cols"])
    try:
        got = oracle.forward(views, pts, meta["patch"], meta["thr"], offs)[1:]
    finally:
        oracle.set_reproject_rule(*prev)
    assert rows_equal(got, ref).all(), "%d rows differ at %d threads" % (int((~rows_equal(got, ref)).sum()), threads)


def test_shell_points_and_volume_multichunk_vs_reference(multichunk):
    """SURVEY.md §8 rows a15-a18 on the four-chunk run: the 9 692 shell points (two of the reference's chunks, PMVO.py:655-691)
    and the voxel fit of surface + shell points (:695-764) -- every row, every voxel, bit for bit."""
    meta, z, views = multichunk
    keep = np.where(z["ref_min_loss"] < meta["threshold"])[0]
    scalp = z["toy_scalp"]
    fu = z["candidates"][z["filter_index"]]
    kept, sori = oracle.shell_orientations(views, z["ref_select_p"][keep], z["ref_select_o"][keep], fu, meta["patch"],
                                           meta["thr"], meta["vis_thr"], KDTree(data=scalp), np.max(scalp, axis=0))
    assert len(fu) == 9692 and np.array_equal(kept, z["ref_filter_unvisible"])
    assert np.array_equal(sori, z["ref_filter_unvisible_ori"])
    sel_o = np.concatenate([z["ref_select_o"][keep], sori], 0)
    sel_p = np.concatenate([z["ref_select_p"][keep], kept], 0)
    occ, ori = oracle.voxel_fit(sel_p.copy(), sel_o.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
    ori_l, occ_l = oracle.mat_layout(occ, ori)
    nz = np.argwhere(occ_l != 0).astype(np.int32)
    assert np.array_equal(nz, z["mat_occ_nz"]) and len(nz) == 23120
    Z = occ_l.shape[2]
    got = np.stack([ori_l[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    assert np.array_equal(got, z["mat_ori_at_nz"])


def test_refine_loop_head_filter_and_nan_rows_vs_reference(multichunk):
    """tests/golden/e2e_headfilter.npz (tools/gen_golden_headfilter.py): the reference's smoothing loop on 6000 points (two
    chunks) of which a third are head-filtered (PMVO.py:91-92: loss -1, then 0.5 at :639) and 40 carry NaN orientations / losses
    that enter their neighbours' medoids as NaN cosines.  Every orientation (NaN == NaN) and every loss outside the per-chunk
    N-mod-64 tails, bit for bit; the shell stage behind it as well."""
    meta, _, views = multichunk
    z = load_npz("e2e_headfilter")
    zz = load_npz("e2e_multichunk")
    scalp = zz["toy_scalp"]
    pts, ori, loss = z["in_points"], z["in_ori"].copy(), z["in_loss"].copy()
    trace = []
    oracle.refine_loop(views, pts, ori, loss, meta["patch"], meta["thr"], meta["vis_thr"], KDTree(data=scalp),
                       np.max(scalp, axis=0), trace=trace)
    ref_o, ref_l = z["ref_select_o"], z["ref_min_loss"]
    assert (ref_l[:5000] == 0.5).sum() == 1667 and (ref_l[5000:] == 0.5).sum() == 333 and np.isnan(ref_l).sum() > 0
    om = np.all((ori == ref_o) | (np.isnan(ori) & np.isnan(ref_o)), axis=1)
    assert om.all(), (float(om.mean()), np.flatnonzero(~om)[:10])
    loss_rows_equal(loss, ref_l)
    assert np.array_equal(loss == 0.5, ref_l == 0.5) and np.array_equal(np.isnan(loss), np.isnan(ref_l))
    keep = np.where(ref_l < meta["threshold"])[0]
    kept, sori = oracle.shell_orientations(views, z["ref_select_p"][keep], ref_o[keep], z["in_shell"], meta["patch"],
                                           meta["thr"], meta["vis_thr"], KDTree(data=scalp), np.max(scalp, axis=0))
    assert np.array_equal(kept, z["ref_filter_unvisible"])
    assert np.array_equal(sori, z["ref_filter_unvisible_ori"], equal_nan=True)


@pytest.mark.parametrize("name", ["pmvo_small", "pmvo_quant"])
def test_batches_of_one_point_vs_reference(name):
    """tests/golden/pmvo_single.npz (tools/gen_golden_single.py): points handed to the reference ALONE -- every projection of the
    point is then a single-column sgemm with its own rounding (oracle/pmvo_oracle.c: batch_is_single), which is what the last chunk
    of optimize / refine is when N mod 5000 == 1.  forward, the method refine and the votes: every row."""
    from conftest import GOLDEN, golden_records, golden_scene, load_golden, scene_views

    meta, z = load_golden(name)
    views = scene_views(golden_scene(meta), golden_records(z))
    s = np.load(__import__("os").path.join(GOLDEN, "pmvo_single.npz"))
    g = lambda k: s[name + "__" + k]                                                    # noqa: E731
    offs = np.load(__import__("os").path.join(GOLDEN, "depth_offsets.npy"))
    differs = 0
    for i, n in enumerate(g("pick")):
        p = z["points"][n:n + 1]
        _, o, l, h = oracle.forward(views, p, meta["patch"], meta["thr"], offs)
        assert np.array_equal(o[0], g("fwd_ori")[i], equal_nan=True) and np.array_equal(l[0], g("fwd_loss")[i], equal_nan=True)
        assert h[0] == g("fwd_hc")[i]
        differs += not np.array_equal(l[0], z["fwd_loss"][n], equal_nan=True)
        rl, _ = oracle.refine_loss(views, p, z["refine_ori_in"][n:n + 1], meta["patch"], meta["thr"])
        _, _, _, head = oracle.filter_votes(views, p, meta["patch"], meta["thr"], meta["vis_thr"])
        want = g("refine_loss")[i]
        if want == -1:
            assert head[0] and not head_top_index(p.astype(np.float32), z["toy_scalp"])[0]
        else:
            assert np.array_equal(rl[0], want, equal_nan=True), (i, rl[0], want)
    assert differs > 5          # alone is not the same as inside the N-point batch: the fixture really pins another form
    for i, n in enumerate(g("fpick")):
        q = z["filter_points_in"][n:n + 1]
        surf, filt, unv, _ = oracle.filter_votes(views, q, meta["patch"], meta["thr"], meta["vis_thr"])
        assert surf[0] == g("surface")[i] and filt[0] == g("filter")[i] and unv[0] == g("unvisible")[i]
