"""GPU: BASELINE.json's full-size configuration (60 views @ 1920x1080, 5000-point iteration, patch 7) checked
through size-independent properties and an oracle comparison on a random subset (points are independent, so
equality on a subset is equality of the path at that size)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def full():
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    V, H, W, patch, thr = 60, 1920, 1080, 7, 0.15
    scene = synth.make_scene(V, H, W, device=DEV, seed=0, quantize=True)   # 8-bit maps: ties, duplicates
    cams = cameras_from_list(scene["cams"])
    rec = camera_records(cams)
    pm = PMVO.from_planes(rec, scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=DEV,
                          patch_size=patch, visible_threshold=1, conf_threshold=thr, camera=cams)
    cand = synth.candidate_points(res=256, seed=0)
    surf, _, _ = pm.filter_points(cand[:60000])
    pts = cand[:60000][surf.cpu().numpy()][:5000]
    assert len(pts) == 5000
    return scene, rec, pm, pts, patch, thr


def test_chunk_matches_oracle_at_full_size(full):
    """one whole 5000-point iteration at the headline size against the oracle on the same batch: every row (the answer of a
    point depends on the batch it is in -- group sizes per base view, trailing columns -- so the comparison is batch to batch)"""
    from monohair_amd.pmvo import depth_offsets

    scene, rec, pm, pts, patch, thr = full
    p, ori, loss, hc, ex = pm.forward(pts, extras=True)
    views = oracle.Views(rec, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(), scene["conf"].cpu().numpy(),
                         scene["mask"].cpu().numpy())
    bidx, bval = ex["base_idx"].cpu().numpy(), ex["base_val"].cpu().numpy()
    _, o_ori, o_loss, o_hc, o_ex = oracle.forward(views, pts, patch, thr, depth_offsets(90), base_idx=bidx, base_val=bval,
                                                  extra=True)
    assert np.array_equal(loss.cpu().numpy(), o_loss, equal_nan=True)
    assert np.array_equal(ori.cpu().numpy(), o_ori, equal_nan=True)
    assert np.array_equal(hc.cpu().numpy(), o_hc)
    assert np.array_equal(ex["best_s"].cpu().numpy(), o_ex["best_s"])
    assert np.isfinite(o_loss).mean() > 0.9
    # this batch exercises every form: groups of one point, mid-size groups, groups past MKL's 316-point switch
    sizes = np.concatenate([oracle.group_sizes(bidx[r], views.V) for r in range(0, 20, 2)])
    assert (sizes == 1).any() and ((sizes > 1) & (sizes <= 316)).any() and (sizes > 316).any(), np.unique(sizes)[[0, -1]]
    # the base-view ranking itself, on a subset
    sel = np.sort(np.random.default_rng(0).choice(5000, 96, replace=False))
    o = oracle.visible_and_ori(views, pts[sel], 1)
    oi, ov = oracle.topk_views(o["visible"], o["Conf"], 20)
    assert np.array_equal(bval[:, sel], ov)


def test_permutation_and_chunk_split_invariance(full):
    """With the batch-independent options (reproject_rule 1, sum_block 0) a point's answer does not depend on what else is in
    the batch: permuting or splitting the chunk changes nothing.  With the defaults (the reference's own batch dependence) a
    permutation keeps every group size, so only the point(s) that enter / leave the trailing columns of the sums can change."""
    scene, rec, pm, pts, patch, thr = full
    perm = np.random.default_rng(1).permutation(5000)
    nn = lambda t: torch.nan_to_num(t, nan=-7.0)                                           # noqa: E731
    pm.set_option("reproject_rule", 1)
    pm.set_option("sum_block", 0)
    try:
        _, ori, loss, hc = pm.forward(pts)
        _, ori_p, loss_p, hc_p = pm.forward(pts[perm])
        assert torch.equal(nn(loss[perm]), nn(loss_p)) and torch.equal(nn(ori[perm]), nn(ori_p)) and torch.equal(hc[perm], hc_p)
        _, ori_a, loss_a, _ = pm.forward(pts[:1777])
        _, ori_b, loss_b, _ = pm.forward(pts[1777:])
        assert torch.equal(nn(torch.cat([loss_a, loss_b])), nn(loss)) and torch.equal(nn(torch.cat([ori_a, ori_b])), nn(ori))
    finally:
        pm.set_option("reproject_rule", 0)
        pm.set_option("sum_block", 32)
    _, ori, loss, hc = pm.forward(pts)
    _, ori_p, loss_p, hc_p = pm.forward(pts[perm])
    free = torch.ones(5000, dtype=torch.bool, device=loss.device)
    free[-1] = False                                  # the last point of the permuted batch (5000 * 90 mod 32 = 16 columns)
    free[int(np.flatnonzero(perm == 4999)[0])] = False   # ... and of the original one
    assert torch.equal(nn(loss[perm])[free], nn(loss_p)[free]) and torch.equal(nn(ori[perm])[free], nn(ori_p)[free])


def test_known_answer_tangent_field(full):
    """the recovered 3D directions follow the meridian tangent field of the sphere (8-bit maps: ~1 degree)"""
    scene, rec, pm, pts, patch, thr = full
    _, ori, loss, hc = pm.forward(pts)
    ori, loss = ori.cpu().numpy(), loss.cpu().numpy()
    n = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    t = -np.array([0, 1.0, 0])[None] + n[:, 1:2] * n
    ok = (np.linalg.norm(t, axis=1) > 0.3) & np.isfinite(loss)
    t = t[ok] / np.linalg.norm(t[ok], axis=1, keepdims=True)
    cosv = np.abs((t * ori[ok]).sum(1))
    assert np.median(cosv) > 0.995, np.median(cosv)
    assert np.all(np.abs(np.linalg.norm(ori[ok], axis=1) - 1) < 1e-5)


def test_streams_do_not_interfere(full):
    """two iterations in flight on two HIP streams (as optimize() issues them) give the serial results"""
    scene, rec, pm, pts, patch, thr = full
    a, b = pts[:2500], pts[2500:]
    _, oa, la, _ = pm.forward(a)
    _, ob, lb, _ = pm.forward(b)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            _, oa2, la2, _ = pm.forward(a)
        with torch.cuda.stream(s2):
            _, ob2, lb2, _ = pm.forward(b)
    torch.cuda.synchronize()
    assert torch.equal(torch.nan_to_num(la, nan=-7.0), torch.nan_to_num(la2, nan=-7.0))
    assert torch.equal(torch.nan_to_num(lb, nan=-7.0), torch.nan_to_num(lb2, nan=-7.0))
    assert torch.equal(torch.nan_to_num(oa, nan=-7.0), torch.nan_to_num(oa2, nan=-7.0))
