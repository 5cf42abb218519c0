"""GPU parity: the HIP path (through the C ABI, via monohair_amd.PMVO) against the CPU oracle on the same
seeded inputs, and against the golden vectors produced by the reference itself.

Bar: bit-exact vs the oracle for every per-(view,point) quantity and for the loss search (the kernels
evaluate the same fp32 operations in the same order); vs the reference goldens the same exceptions apply as
for the oracle itself (tests/test_oracle_golden.py)."""
import numpy as np
import pytest
import torch

import oracle
from conftest import golden_records, golden_scene, load_golden, scene_views

pytestmark = pytest.mark.gpu

CASES = ["pmvo_small", "pmvo_mid", "pmvo_quant", "pmvo_views300", "pmvo_views300c", "pmvo_patch9", "pmvo_patch4"]


def eq_nan(a, b):
    return np.array_equal(a, b, equal_nan=True)


def make_pmvo(meta, scene, records):
    """PMVO on device-resident planes with the golden's camera records (see conftest.golden_records)."""
    from monohair_amd.pmvo import PMVO

    dev = "cuda:0"
    return PMVO.from_planes(records, scene["depth"].to(dev), scene["ori"].to(dev), scene["conf"].to(dev),
                            scene["mask"].to(dev), device=dev, patch_size=meta["patch"],
                            visible_threshold=meta["vis_thr"], conf_threshold=meta["thr"])


@pytest.fixture(scope="module", params=CASES)
def case(request):
    meta, z = load_golden(request.param)
    scene = golden_scene(meta)
    rec = golden_records(z)
    return meta, z, scene, scene_views(scene, rec), make_pmvo(meta, scene, rec)


def test_reference_style_constructor(case):
    """PMVO(camera, depths, Ori, Conf, masks, ...) with the reference's dict-of-numpy arguments
    (PMVO.py:14-28) packs the same maps as the device-plane constructor."""
    from monohair_amd import synth
    from monohair_amd.camera import cameras_from_list
    from monohair_amd.pmvo import PMVO

    meta, z, scene, views, pm = case
    cams = cameras_from_list(scene["cams"])
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
    pm2 = PMVO(cams, depths, Ori, Conf, masks, device="cuda:0", image_size=[meta["H"], meta["W"]],
               patch_size=meta["patch"], visible_threshold=meta["vis_thr"], conf_threshold=meta["thr"])
    assert pm2.camera_key == [c["file"] for c in scene["cams"]]
    pm.Compute_Visible_and_Ori(z["points"])
    pm2.Compute_Visible_and_Ori(z["points"])
    for k in ("visible", "Ori", "Conf", "mask", "Ori_patch", "Conf_patch"):
        assert torch.equal(getattr(pm, k), getattr(pm2, k)), k


def test_depth_offsets_match_fixture(depth_offsets):
    from monohair_amd.pmvo import depth_offsets as mk

    assert np.array_equal(mk(90), depth_offsets)


def test_project_gather_vs_oracle_and_golden(case):
    meta, z, scene, views, pm = case
    pm.Compute_Visible_and_Ori(z["points"])
    o = oracle.visible_and_ori(views, z["points"], meta["patch"])
    got = dict(visible=pm.visible, Ori=pm.Ori, Conf=pm.Conf, mask=pm.mask, Ori_patch=pm.Ori_patch,
               Conf_patch=pm.Conf_patch, pixf=pm._pixf)
    for k, t in got.items():
        assert np.array_equal(t.cpu().numpy(), o[k]), k
    for k in ("visible", "Ori", "Conf", "mask"):
        assert np.array_equal(got[k].cpu().numpy(), z[k]), k
    nd = meta["n_d"]
    assert np.array_equal(pm.Ori_patch[:, :nd].cpu().numpy(), z["Ori_patch_head"])
    assert np.array_equal(pm.Conf_patch[:, :nd].cpu().numpy(), z["Conf_patch_head"])


def test_topk_vs_oracle(case):
    meta, z, scene, views, pm = case
    pm.Compute_Visible_and_Ori(z["points"])
    idx, val = pm.Find_max_conf_from_visible_view()
    oi, ov = oracle.topk_views(z["visible"], z["Conf"], 20)
    assert idx.dtype == torch.int64
    assert np.array_equal(idx.cpu().numpy(), oi)
    assert np.array_equal(val.cpu().numpy(), ov)
    # ... which is the reference's own torch.topk output, equal values included (csrc/mh_topk_order.h)
    assert np.array_equal(val.cpu().numpy(), z["base_val"]) and np.array_equal(idx.cpu().numpy(), z["base_idx"])
    pm.set_option("topk_order", 1)                      # round 1's rule: value descending, view index ascending
    idx1, val1 = pm.Find_max_conf_from_visible_view()
    pm.set_option("topk_order", 0)
    o1, v1 = oracle.topk_views(z["visible"], z["Conf"], 20, order="index")
    assert np.array_equal(idx1.cpu().numpy(), o1) and np.array_equal(val1.cpu().numpy(), v1)


@pytest.mark.parametrize("variant", [0, 7, 100, 107, 1256])   # shipped key body (work-ordered / natural order), the
# compare-and-select body of rounds 1-3 in both orders, portable cross-check
def test_forward_vs_oracle_bit_exact(case, depth_offsets, variant):
    meta, z, scene, views, pm = case
    pm.set_option("search_variant", variant)
    pts = z["points"]
    p, ori, loss, hc, ex = pm.forward(pts, base_view=(z["base_idx"], z["base_val"]), extras=True)
    _, o_ori, o_loss, o_hc, o_ex = oracle.forward(views, pts, meta["patch"], meta["thr"], depth_offsets,
                                                  base_idx=z["base_idx"], base_val=z["base_val"], extra=True)
    pm.set_option("search_variant", 0)
    assert eq_nan(loss.cpu().numpy(), o_loss)
    assert np.array_equal(ex["best_rank"].cpu().numpy(), o_ex["best_rank"])
    assert np.array_equal(ex["best_s"].cpu().numpy(), o_ex["best_s"])
    assert eq_nan(ex["best_sample"].cpu().numpy(), o_ex["best_sample"])
    assert eq_nan(ori.cpu().numpy(), o_ori)
    assert np.array_equal(hc.cpu().numpy(), o_hc)
    assert hc.dtype == torch.bool and ori.dtype == torch.float32 and loss.dtype == torch.float32


def test_forward_vs_reference_golden(case, depth_offsets, request):
    """Against the reference's own outputs, EVERY row, bit for bit, in every batch composition the reference was run in
    (tools/gen_golden_recompose.py): the original batch, the batch reversed, the batch doubled.  The reference's answer for a
    point depends on its batch (MKL's sgemm kernel by the number of points that share a base view, ATen's trailing columns);
    the kernels follow the batch (options reproject_rule 0, sum_block 32: include/mh_pmvo.h)."""
    from conftest import recompose_golden, rows_equal

    meta, z, scene, views, pm = case
    name = request.node.callspec.params["case"]
    pts = z["points"]
    N = len(pts)
    for variant in (0, 1256):
        pm.set_option("search_variant", variant)
        fwd = lambda p, **kw: tuple(t.cpu().numpy() for t in pm.forward(p, **kw)[1:])      # noqa: E731
        assert rows_equal(fwd(pts, base_view=(z["base_idx"], z["base_val"])), (z["fwd_ori"], z["fwd_loss"], z["fwd_hc"])).all()
        assert rows_equal(tuple(a[::-1] for a in fwd(pts[::-1].copy())), recompose_golden(name, "rev")).all()
        assert rows_equal(tuple(a[:N] for a in fwd(np.concatenate([pts, pts], 0))), recompose_golden(name, "dup")).all()
    pm.set_option("search_variant", 0)


def test_forward_forced_mid_forms_vs_reference_golden(case, depth_offsets, request):
    """The batch-independent options (reproject_rule 1, sum_block 0: what rounds 1-4 computed): equal to the oracle under the
    same options, and to the reference's doubled-batch answer on every row; rows that differ from its original-batch answer
    are rows on which the reference disagrees with itself (conftest.check_forward_against_reference)."""
    from conftest import check_forward_against_reference

    meta, z, scene, views, pm = case
    pts = z["points"]
    pm.set_option("reproject_rule", 1)
    pm.set_option("sum_block", 0)
    prev = oracle.set_reproject_rule("mid"), oracle.set_sum_block(0)
    try:
        p, ori, loss, hc = pm.forward(pts, base_view=(z["base_idx"], z["base_val"]))
        _, o_ori, o_loss, o_hc = oracle.forward(views, pts, meta["patch"], meta["thr"], depth_offsets,
                                                base_idx=z["base_idx"], base_val=z["base_val"])
    finally:
        pm.set_option("reproject_rule", 0)
        pm.set_option("sum_block", 32)
        oracle.set_reproject_rule(*prev[0])
        oracle.set_sum_block(prev[1])
    assert eq_nan(loss.cpu().numpy(), o_loss) and eq_nan(ori.cpu().numpy(), o_ori) and np.array_equal(hc.cpu().numpy(), o_hc)
    check_forward_against_reference(request.node.callspec.params["case"], z, ori.cpu().numpy(), loss.cpu().numpy(),
                                    hc.cpu().numpy())


def test_fused_and_unfused_front_ends_agree(case):
    """forward(fused=True) (projection + tap lists straight from the maps, invisible views never gathered) and
    forward(fused=False) (materialised patch tensors) are the same computation; the patch attributes appear lazily."""
    meta, z, scene, views, pm = case
    pts = z["points"]
    _, o1, l1, h1 = pm.forward(pts, fused=True)
    vis1, ori1, conf1, mask1 = pm.visible.clone(), pm.Ori.clone(), pm.Conf.clone(), pm.mask.clone()
    lazy = pm.Ori_patch.clone()          # materialised on first access
    _, o2, l2, h2 = pm.forward(pts, fused=False)
    for a, b in ((o1, o2), (l1, l2)):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    assert torch.equal(h1, h2)
    assert torch.equal(vis1, pm.visible) and torch.equal(ori1, pm.Ori) and torch.equal(conf1, pm.Conf)
    assert torch.equal(mask1, pm.mask) and torch.equal(lazy, pm.Ori_patch)


def test_tap_plane_on_and_off_agree(case):
    """option "tap_plane": the fused front end gathers ready-made taps (normalised and clamped once, at upload) or normalises
    per iteration -- the same tap lists, the same results, bit for bit"""
    meta, z, scene, views, pm = case
    pts = z["points"]
    res = {}
    for use in (1, 0):
        pm.set_option("tap_plane", use)
        _, o, l, h = pm.forward(pts)
        res[use] = (o.clone(), l.clone(), h.clone(), pm.search_work(len(pts))[0].clone(), pm.visible.clone(), pm.Conf.clone())
    pm.set_option("tap_plane", 1)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b.float(), nan=-7.0))


def test_forward_own_ranking_runs_and_matches_oracle(case, depth_offsets):
    meta, z, scene, views, pm = case
    pts = z["points"]
    p, ori, loss, hc = pm.forward(pts)
    _, o_ori, o_loss, o_hc = oracle.forward(views, pts, meta["patch"], meta["thr"], depth_offsets)
    assert eq_nan(loss.cpu().numpy(), o_loss)
    assert eq_nan(ori.cpu().numpy(), o_ori)
    assert np.array_equal(hc.cpu().numpy(), o_hc)
    assert torch.equal(p.cpu(), torch.from_numpy(pts).float())


def test_empty_and_ragged_batches(case, depth_offsets):
    meta, z, scene, views, pm = case
    p, ori, loss, hc = pm.forward(np.zeros((0, 3)))
    assert ori.shape == (0, 3) and loss.shape == (0,) and hc.shape == (0,)
    # a batch that is not a multiple of the 64-point tile, including a point far away from the object
    pts = np.concatenate([z["points"][:67], np.array([[5.0, 5.0, 5.0]])], 0)
    p, ori, loss, hc = pm.forward(pts)
    assert ori.shape == (len(pts), 3)
    _, o_ori, o_loss, o_hc = oracle.forward(views, pts, meta["patch"], meta["thr"], depth_offsets)
    assert eq_nan(loss.cpu().numpy(), o_loss) and eq_nan(ori.cpu().numpy(), o_ori)
    if meta["rings"] == 1:                     # a single ring of cameras: the far point is outside every frustum
        assert np.isnan(loss[-1].item())


def test_intermediate_methods_vs_oracle_and_golden(case, depth_offsets):
    """The reference's helper methods (project_points, get_*, compute_visible, sample_next_3d_pos,
    compute_reproject_ori, compute_points_prj_ori, compute_prj_loss) as stand-alone calls: same results as the oracle's
    restatements bit for bit, and the reference's own outputs where the goldens hold them."""
    meta, z, scene, views, pm = case
    pts = z["points"]
    H, W, patch = meta["H"], meta["W"], meta["patch"]
    # project_points + get_* on two views
    for tag in ("a", "b"):
        v = int(z["proj_%s_view" % tag])
        uv, zp, oob = pm.project_points(pts, pm.camera_key[v] if pm.camera_key else v, [H, W])
        o_rc, o_zp, o_oob, _ = oracle.project_points(views.cams[v], pts, H, W)
        assert np.array_equal(uv.cpu().numpy(), o_rc) and np.array_equal(zp.cpu().numpy(), o_zp)
        assert np.array_equal(oob.cpu().numpy(), o_oob)
        assert np.array_equal(uv.cpu().numpy(), z["proj_%s_rc" % tag]) and np.array_equal(oob.cpu().numpy(), z["proj_%s_oob" % tag])
        assert np.array_equal(zp.cpu().numpy(), z["proj_%s_z" % tag])
        r, c = o_rc[:, 0], o_rc[:, 1]
        assert np.array_equal(pm.get_depth(uv, v).cpu().numpy(), views.depth[v][r, c])
        assert np.array_equal(pm.get_ori(uv, v).cpu().numpy(), views.ori[v][r, c])
        assert np.array_equal(pm.get_conf(uv, v).cpu().numpy(), views.conf[v][r, c])
        assert np.array_equal(pm.get_mask(uv, v).cpu().numpy(), views.mask[v][r, c])
        hp = patch // 2
        taps = [(np.clip(r + i, 0, H - 1), np.clip(c + j, 0, W - 1)) for i in range(-hp, hp + 1) for j in range(-hp, hp + 1)]
        assert np.array_equal(pm.get_ori_patch(uv, v, patch).cpu().numpy(), np.stack([views.ori[v][a, b] for a, b in taps], 1))
        assert np.array_equal(pm.get_c_patch(uv, v, patch).cpu().numpy(), np.stack([views.conf[v][a, b] for a, b in taps], 1))
        vis = pm.compute_visible(pm.get_depth(uv, v), zp * 255)
        assert np.array_equal(np.where(o_oob, -1.0, vis.cpu().numpy()).astype(np.float32), z["visible"][v])
    # the search, step by step, against the fused forward's ingredients
    pm.Compute_Visible_and_Ori(pts)
    for rank in (0, 2):
        base = z["base_idx"][rank]
        samples, surface = pm.sample_next_3d_pos(pts, base)
        o_s = oracle.sample_next(views, pts, base, z["Ori"], depth_offsets)
        assert np.array_equal(samples.cpu().numpy(), o_s) and torch.equal(surface.cpu(), torch.from_numpy(pts).float())
        assert np.array_equal(o_s, z["samples_r%d" % rank])           # == the reference's own batch, every sample
        D = pm.compute_reproject_ori(pts, samples)
        o_D = oracle.reproject_ori(views, pts, o_s)
        assert eq_nan(D.cpu().numpy(), o_D)
        one = pm.compute_points_prj_ori(pts, samples[:, 7])
        assert eq_nan(one.cpu().numpy(), o_D[:, :, 7])
        loss, idx, hc = pm.compute_prj_loss(D, None, None)
        o = oracle.visible_and_ori(views, pts, patch)
        o_loss, o_idx, o_hc = oracle.prj_loss(o_D, o["Ori_patch"], o["Conf_patch"], o["visible"], meta["thr"])
        assert eq_nan(loss.cpu().numpy(), o_loss) and np.array_equal(idx.cpu().numpy(), o_idx)
        assert np.array_equal(hc.cpu().numpy(), o_hc)
        assert eq_nan(o_loss, z["loss_r%d" % rank]) and np.array_equal(o_idx, z["idx_r%d" % rank])   # the reference, every row
        w = pm.compute_weight(pm.visible, pm.Conf, pm.mask)
        assert torch.equal(w, (pm.visible != -1).float() * pm.Conf)


@pytest.mark.parametrize("V,patch", [(20, 7), (23, 7), (27, 9), (33, 5), (41, 11), (58, 7)])
def test_item_slices_both_bodies(V, patch, depth_offsets):
    """How many of the 10 base-view ranks of a point are usable decides how its 90-sample runs fall on the 64-lane item
    slices of the four waves (10 ranks: 4 + 4 + 4 + 3 slices, the last one of 4 items; 5 ranks: 2 + 2 + 2 + 2 with a last
    slice of 2; ...): every count of slices per wave is its own instantiation of the tap loop (hand-ordered key blocks for
    4, 3 and 2 items, the generic form for 1).  Camera counts from 20 up give points with every count; patch 9 and 11 give tap
    lists of two 64-tap groups (their own kernel instantiation).  Key body, select body and the portable kernel against the oracle, every point."""
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    DEV = "cuda:0"
    H, W, thr = 200, 150, 0.15
    scene = synth.make_scene(V, H, W, device=DEV, seed=100 + V, quantize=False)
    cams = cameras_from_list(scene["cams"])
    rec = camera_records(cams)
    pm = PMVO.from_planes(rec, scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=DEV, patch_size=patch,
                          visible_threshold=1, conf_threshold=thr, camera=cams)
    views = oracle.Views(rec, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(), scene["conf"].cpu().numpy(),
                         scene["mask"].cpu().numpy())
    pts = synth.candidate_points(res=64, seed=V, limit=700)
    _, o_ori, o_loss, o_hc, o_ex = oracle.forward(views, pts, patch, thr, depth_offsets, extra=True)
    usable = set()
    for body, variant in ((1, 0), (2, 0), (0, 1256)):
        pm.set_option("search_body", body)
        pm.set_option("search_variant", variant)
        _, ori, loss, hc, ex = pm.forward(pts, extras=True)
        assert eq_nan(loss.cpu().numpy(), o_loss), (body, variant)
        assert eq_nan(ori.cpu().numpy(), o_ori) and np.array_equal(hc.cpu().numpy(), o_hc), (body, variant)
        assert np.array_equal(ex["best_s"].cpu().numpy(), o_ex["best_s"]), (body, variant)
        assert np.array_equal(ex["best_rank"].cpu().numpy(), o_ex["best_rank"]), (body, variant)
    pm.set_option("search_body", 0)
    pm.set_option("search_variant", 0)
    # the scene really has points with different numbers of usable ranks (rank r > 0 is usable if base_view_conf[2r] > 0)
    _, val = pm.Find_max_conf_from_visible_view()
    nvalid = 1 + (val.cpu().numpy()[2:20:2] > 0).sum(0)
    usable.update(np.unique(nvalid).tolist())
    assert len(usable) >= 3, usable
