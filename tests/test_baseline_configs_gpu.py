"""GPU: the other configurations BASELINE.json lists, as parity cases (bench.py measures configs[2] only):
  configs[0]  big_wavy1 parameters at 512x512 / 64^3   -- with 20 views: the reference cannot run 8 (PMVO.py:341)
  configs[1]  big_wavy1 parameters, 30 views @ 1080p / 128^3
  configs[4]  120 views @ 3840x2160 / 512^3, Gabor bank as the FP32-MFMA im2col contraction
Points are independent of each other, so bit-equality with the oracle on a random subset of the candidates is
equality of the path at that size.  big_wavy1.yaml:16-20: patch 7, conf_threshold 0.15."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PATCH, THR = 7, 0.15


def eq(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("V,H,W,res,quant", [(20, 512, 512, 64, True), (30, 1920, 1080, 128, False),
                                              (60, 1920, 1080, 256, False),        # bench.py's own scene (continuous maps)
                                              (120, 3840, 2160, 512, True)])
def test_config_subset_parity(V, H, W, res, quant):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO, depth_offsets

    scene = synth.make_scene(V, H, W, device=DEV, seed=1, quantize=quant)
    cams = cameras_from_list(scene["cams"])
    rec = camera_records(cams)
    pm = PMVO.from_planes(rec, scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=DEV, patch_size=PATCH,
                          visible_threshold=1, conf_threshold=THR, camera=cams)
    cand = synth.candidate_points(res=min(res, 256), seed=2)          # same shell, the grid only sets the count
    rng = np.random.default_rng(V)
    pts = cand[rng.choice(len(cand), 400, replace=False)]
    views = oracle.Views(rec, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(), scene["conf"].cpu().numpy(),
                         scene["mask"].cpu().numpy())
    surf, _, filt = pm.filter_points(pts)
    unv = pm.compute_unvisible_points(pts)
    o_s, o_f, o_u, _ = oracle.filter_votes(views, pts, PATCH, THR, 1.0)
    assert eq(surf.cpu().numpy(), o_s) and eq(filt.cpu().numpy(), o_f) and eq(unv.cpu().numpy(), o_u)
    assert 0.2 < o_s.mean() < 1.0
    for fused in (True, False):
        _, ori, loss, hc = pm.forward(pts, fused=fused)
        _, o_ori, o_loss, o_hc = oracle.forward(views, pts, PATCH, THR, depth_offsets(90))
        assert eq(loss.cpu().numpy(), o_loss) and eq(ori.cpu().numpy(), o_ori) and eq(hc.cpu().numpy(), o_hc)
    # known answer: on visible surface points the found direction is the meridian tangent
    keep = o_s & o_hc & np.isfinite(o_loss)
    p = pts[keep]
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    t = -np.array([0, 1.0, 0])[None] + n[:, 1:2] * n
    ok = np.linalg.norm(t, axis=1) > 0.3
    t = t[ok] / np.linalg.norm(t[ok], axis=1, keepdims=True)
    cosv = np.abs((t * o_ori[keep][ok]).sum(1))
    assert np.median(cosv) > 0.95, np.median(cosv)


@pytest.mark.parametrize("H,W", [(1920, 1080), (3840, 2160)])
def test_gabor_shipped_kernel_at_full_size(H, W):
    """configs[2]/[4]: the shipped bank kernel (mfma2: the 180x289 bank as an im2col contraction on the FP32 matrix cores,
    GaborFilter.py:29-113) at 1080p and 3840x2160 gives the bits of the VALU cross-check kernel on every pixel, and both those
    of the oracle on crops (index and un-normalised variance are local, so pixels further than the 8-pixel filter radius from
    an interior crop border do not see the crop; crops that touch the image border share its zero padding there)."""
    from monohair_amd.gabor import calOrientationGabor, gabor_bank

    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    img = (0.25 * torch.cos(2 * np.pi * (0.6 * xx + 0.8 * yy) / 4.0) + 0.02 * torch.randn((H, W), generator=g)).float()
    outs = {}
    for variant in ("mfma2", "valu"):
        gf = calOrientationGabor(device=DEV, variant=variant)
        idx, conf, var = gf.filter_index(img.to(DEV))
        outs[variant] = (idx.cpu().numpy(), conf.cpu().numpy(), var.cpu().numpy())
    assert calOrientationGabor(device=DEV).variant == "mfma2"
    for a, b in zip(outs["mfma2"], outs["valu"]):
        assert np.array_equal(a, b)
    n = 160
    bank = gabor_bank()
    for r0, c0 in ((H // 2 - 100, W // 2 - 80), (0, 0), (H - n, W - n), (0, W - n), (H // 3, 0)):
        o_idx, _, o_var = oracle.gabor_bank(bank, img[r0:r0 + n, c0:c0 + n].numpy())
        rs = slice(0 if r0 == 0 else 8, n if r0 + n == H else n - 8)
        cs = slice(0 if c0 == 0 else 8, n if c0 + n == W else n - 8)
        assert np.array_equal(outs["mfma2"][0][r0:r0 + n, c0:c0 + n][rs, cs], o_idx[rs, cs]), (r0, c0)
        assert np.array_equal(outs["mfma2"][2][r0:r0 + n, c0:c0 + n][rs, cs], o_var[rs, cs]), (r0, c0)
