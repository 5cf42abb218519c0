"""Maps uploaded as 8-bit file codes keep the codes resident (2 B per pixel) and the fused front end of forward() gathers
them instead of the decoded 16-byte records (mh_project_taps_kernel<.., CODES>).  Checked here: forward() is bit-identical
with the option on and off, for every patch size, on the reference's 8-bit fixtures (against the oracle AND the reference's
golden outputs) and on random code maps incl. orientation codes > 180 (the loaders' uint8 wrap) and saturated confidences;
the tap lists of the code form are the fp32 form's lists minus duplicates; mixing upload forms falls back to the records."""
import numpy as np
import pytest
import torch

import oracle
from conftest import golden_records, golden_scene, load_golden, scene_views

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _codes_from_scene(scene):
    """8-bit codes whose decoded records equal the (quantised) fp32 planes of a golden scene"""
    from monohair_amd.pmvo_utils import map_code_lut

    lut = map_code_lut()
    ori, conf, mask = scene["ori"].numpy(), scene["conf"].numpy(), scene["mask"].numpy()
    # invert the table on the values that occur (exact float equality: the fixtures were made through the same table)
    k8 = np.zeros(conf.shape, np.uint8)
    found = np.zeros(conf.shape, bool)
    for c in range(181):
        hit = (ori[..., 0] == lut[c, 0]) & (ori[..., 1] == lut[c, 1]) & ~found
        k8[hit] = c
        found |= hit
    c8 = np.rint(conf.astype(np.float64) * 255).astype(np.uint8)
    m8 = np.where(mask > 0, 255, 0).astype(np.uint8)
    ok = found.all() and np.array_equal(lut[c8, 2], conf) and np.array_equal(lut[m8, 3], mask)
    return k8, c8, m8, ok


@pytest.mark.parametrize("name", ["pmvo_quant", "pmvo_patch9"])
def test_forward_from_codes_equals_oracle_and_reference(name, depth_offsets):
    from conftest import rows_equal
    from monohair_amd.camera import cameras_from_list
    from monohair_amd.pmvo import PMVO

    meta, z = load_golden(name)
    scene = golden_scene(meta)
    k8, c8, m8, ok = _codes_from_scene(scene)
    assert ok, "the fixture's maps are not on the 8-bit grid"
    cams = cameras_from_list(scene["cams"])
    pm = PMVO.from_u8(cams, scene["depth"].numpy(), k8, c8, m8, device=DEV, image_size=[meta["H"], meta["W"]],
                      patch_size=meta["patch"], visible_threshold=meta["vis_thr"], conf_threshold=meta["thr"],
                      records=golden_records(z))
    views = scene_views(scene, golden_records(z))
    pts = z["points"]
    outs = []
    for use in (1, 0):
        pm.set_option("tap_codes", use)
        _, ori, loss, hc = pm.forward(pts, base_view=(z["base_idx"], z["base_val"]))
        outs.append((ori.cpu().numpy(), loss.cpu().numpy(), hc.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b, equal_nan=True)
    _, o_ori, o_loss, o_hc = oracle.forward(views, pts, meta["patch"], meta["thr"], depth_offsets, base_idx=z["base_idx"],
                                            base_val=z["base_val"])
    assert np.array_equal(outs[0][1], o_loss, equal_nan=True) and np.array_equal(outs[0][0], o_ori, equal_nan=True)
    assert np.array_equal(outs[0][2], o_hc)
    assert rows_equal(outs[0], (z["fwd_ori"], z["fwd_loss"], z["fwd_hc"])).all()     # the reference's own batch, every row


@pytest.mark.parametrize("patch", [1, 3, 5, 7, 9, 11])
def test_random_code_maps_all_patch_sizes(patch, depth_offsets):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO
    from monohair_amd.pmvo_utils import map_code_lut

    V, H, W = 22, 120, 88
    sc = synth.make_scene_codes(V, H, W, seed=patch)
    rng = np.random.default_rng(patch)
    k8 = sc["ori_u8"].numpy().copy()
    c8 = sc["conf_u8"].numpy().copy()
    # noise on the orientation codes (many distinct codes per patch, some > 180: the loaders' uint8 wrap; 0 / 180 decode to
    # the same line direction with different bits) and confidences around the threshold
    flip = rng.random(k8.shape) < 0.3
    k8[flip] = rng.integers(0, 256, size=int(flip.sum())).astype(np.uint8)
    c8[rng.random(c8.shape) < 0.2] = rng.integers(0, 80, size=1).astype(np.uint8)[0]
    cams = cameras_from_list(sc["cams"])
    thr = 0.15
    pm = PMVO.from_u8(cams, sc["depth"].numpy(), k8, c8, sc["mask_u8"].numpy(), device=DEV, image_size=[H, W],
                      patch_size=patch, visible_threshold=1, conf_threshold=thr)
    lut = map_code_lut()
    views = oracle.Views(camera_records(cams), sc["depth"].numpy(), lut[k8][..., :2].copy(), lut[c8][..., 2].copy(),
                         lut[sc["mask_u8"].numpy()][..., 3].copy())
    pts = synth.candidate_points(res=32, seed=3, limit=260)
    res = {}
    for use in (1, 0):
        pm.set_option("tap_codes", use)
        _, ori, loss, hc = pm.forward(pts)
        res[use] = (ori.cpu().numpy(), loss.cpu().numpy(), hc.cpu().numpy(), pm.search_work(len(pts))[0].cpu().numpy())
    for a, b in zip(res[1][:3], res[0][:3]):
        assert np.array_equal(a, b, equal_nan=True)
    assert (res[1][3] <= res[0][3]).all()                 # the code form drops every duplicate, the hash form nearly every
    assert (res[1][3] > 0).sum() == (res[0][3] > 0).sum()
    _, o_ori, o_loss, o_hc = oracle.forward(views, pts, patch, thr, depth_offsets)
    assert np.array_equal(res[1][1], o_loss, equal_nan=True) and np.array_equal(res[1][0], o_ori, equal_nan=True)
    assert np.array_equal(res[1][2], o_hc)
    # both tap bodies of the search (a context of 8-bit views takes the select body by default; the key body on these noisy
    # code maps sees lists of up to patch^2 taps -- one or two 64-tap groups -- whose losses tie all the time)
    for body in (1, 2):
        pm.set_option("search_body", body)
        _, ori, loss, hc = pm.forward(pts)
        assert np.array_equal(loss.cpu().numpy(), o_loss, equal_nan=True), body
        assert np.array_equal(ori.cpu().numpy(), o_ori, equal_nan=True) and np.array_equal(hc.cpu().numpy(), o_hc), body
    pm.set_option("search_body", 0)


def test_mixed_upload_forms_fall_back_to_the_records(depth_offsets):
    """a context whose views were not ALL uploaded as codes gathers the fp32 records (same results)"""
    from monohair_amd import synth
    from monohair_amd.camera import cameras_from_list
    from monohair_amd.pmvo import PMVO

    V, H, W = 20, 96, 64
    sc = synth.make_scene_codes(V, H, W, seed=4)
    cams = cameras_from_list(sc["cams"])
    pm = PMVO.from_u8(cams, sc["depth"].numpy(), sc["ori_u8"].numpy(), sc["conf_u8"].numpy(), sc["mask_u8"].numpy(),
                      device=DEV, image_size=[H, W], patch_size=5, visible_threshold=1, conf_threshold=0.15)
    pts = synth.candidate_points(res=32, seed=5, limit=200)
    _, o1, l1, h1 = pm.forward(pts)
    n_codes = pm.search_work(len(pts))[0].clone()
    # view 3 again, as decoded float planes (mh_ctx_set_view): its codes are no longer resident
    from monohair_amd import _lib
    from monohair_amd.camera import camera_records
    from monohair_amd.pmvo_utils import map_code_lut

    lut = map_code_lut()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)       # noqa: E731
    k, c, m = (sc[n][3].numpy() for n in ("ori_u8", "conf_u8", "mask_u8"))
    pm._set_view(3, camera_records(cams)[3], t(sc["depth"][3].numpy()), 1, t(lut[k][..., :2]), t(lut[c][..., 2]),
                 t(lut[m][..., 3]), 1, _lib.stream_ptr())
    _, o2, l2, h2 = pm.forward(pts)
    assert torch.equal(torch.nan_to_num(o1, nan=-7.0), torch.nan_to_num(o2, nan=-7.0))
    assert torch.equal(torch.nan_to_num(l1, nan=-7.0), torch.nan_to_num(l2, nan=-7.0)) and torch.equal(h1, h2)
    assert (pm.search_work(len(pts))[0] >= n_codes).all()
