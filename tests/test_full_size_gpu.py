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


def test_subset_matches_oracle_at_full_size(full):
    from monohair_amd.pmvo import depth_offsets

    scene, rec, pm, pts, patch, thr = full
    p, ori, loss, hc, ex = pm.forward(pts, extras=True)
    sel = np.sort(np.random.default_rng(0).choice(5000, 96, replace=False))
    views = oracle.Views(rec, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(), scene["conf"].cpu().numpy(),
                         scene["mask"].cpu().numpy())
    bidx = ex["base_idx"].cpu().numpy()[:, sel]
    bval = ex["base_val"].cpu().numpy()[:, sel]
    _, o_ori, o_loss, o_hc, o_ex = oracle.forward(views, pts[sel], patch, thr, depth_offsets(90), base_idx=bidx,
                                                  base_val=bval, extra=True)
    assert np.array_equal(loss.cpu().numpy()[sel], o_loss, equal_nan=True)
    assert np.array_equal(ori.cpu().numpy()[sel], o_ori, equal_nan=True)
    assert np.array_equal(hc.cpu().numpy()[sel], o_hc)
    assert np.array_equal(ex["best_s"].cpu().numpy()[sel], o_ex["best_s"])
    assert np.isfinite(o_loss).mean() > 0.9
    # the base-view ranking itself, on the subset
    o = oracle.visible_and_ori(views, pts[sel], 1)
    oi, ov = oracle.topk_views(o["visible"], o["Conf"], 20)
    assert np.array_equal(bval, ov)


def test_permutation_and_chunk_split_invariance(full):
    scene, rec, pm, pts, patch, thr = full
    _, ori, loss, hc = pm.forward(pts)
    perm = np.random.default_rng(1).permutation(5000)
    _, ori_p, loss_p, hc_p = pm.forward(pts[perm])
    assert torch.equal(torch.nan_to_num(loss[perm], nan=-7.0), torch.nan_to_num(loss_p, nan=-7.0))
    assert torch.equal(torch.nan_to_num(ori[perm], nan=-7.0), torch.nan_to_num(ori_p, nan=-7.0))
    assert torch.equal(hc[perm], hc_p)
    _, ori_a, loss_a, _ = pm.forward(pts[:1777])
    _, ori_b, loss_b, _ = pm.forward(pts[1777:])
    assert torch.equal(torch.nan_to_num(torch.cat([loss_a, loss_b]), nan=-7.0), torch.nan_to_num(loss, nan=-7.0))
    assert torch.equal(torch.nan_to_num(torch.cat([ori_a, ori_b]), nan=-7.0), torch.nan_to_num(ori, nan=-7.0))


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
