"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF
(imported read-only from /root/reference, CPU, torch) on seeded synthetic scenes.

Runs only in the build container (the reference does not exist on the GPU box).
The fixtures are data: scene parameters, input points and the reference's outputs.
Maps are not stored -- monohair_amd.synth regenerates them bit-identically from the
stored parameters (IEEE basic ops only); a checksum of every map is stored to prove it.

    python tools/gen_golden.py [--only NAME]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)

from ref_import import import_reference  # noqa: E402
from monohair_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

PMVO_CASES = {
    # name: scene + method parameters
    "pmvo_small": dict(V=24, H=96, W=64, seed=3, scale=1.7, rings=1, quantize=False, res=32, N=160, patch=3,
                       thr=0.15, vis_thr=1.0, pt_seed=11, n_d=24),
    "pmvo_mid": dict(V=24, H=480, W=270, seed=0, scale=1.7, rings=2, quantize=False, res=64, N=240, patch=7,
                     thr=0.15, vis_thr=1.0, pt_seed=5, n_d=12),
    "pmvo_quant": dict(V=30, H=240, W=136, seed=7, scale=1.7, rings=1, quantize=True, res=64, N=200, patch=5,
                       thr=0.1, vis_thr=1.0, pt_seed=9, n_d=12),
    # more than 256 views (a real capture of ~300 frames is used unstrided, Camera_utils.py:148-163): ATen's cascade
    # sum moves to a third level every 256 rows
    # (this case stores its scene: with so many camera angles the host's libm shows up in the last bit of a few map
    # values, and the GPU tests run on another CPU than the one the goldens come from)
    "pmvo_views300": dict(V=300, H=40, W=32, seed=2, scale=1.7, rings=3, quantize=False, res=32, N=48, patch=3,
                          thr=0.15, vis_thr=1.0, pt_seed=13, n_d=8, store_scene=True),
    # the same 300 views with the points in tight clusters of four: every base view then owns >= 2 points at every rank,
    # so the reference's 3x3 product in Camera.reprojection takes MKL's gemm kernel (the one the oracle restates) instead
    # of the gemv path a single-point view lands in -- this case can be held to the thresholds of the small ones
    "pmvo_views300c": dict(V=300, H=40, W=32, seed=2, scale=1.7, rings=3, quantize=False, res=32, N=48, patch=3,
                           thr=0.15, vis_thr=1.0, pt_seed=14, n_d=8, store_scene=True, cluster=4),
    # the other patch sizes: 9 x 9 taps on 8-bit maps with the minimum number of views the reference accepts (20), and an
    # EVEN patch size (range(-(4//2), 4//2+1) is the 5 x 5 window, PMVO.py:494-495) with another confidence threshold
    "pmvo_patch9": dict(V=20, H=160, W=120, seed=5, scale=1.7, rings=1, quantize=True, res=48, N=96, patch=9,
                        thr=0.2, vis_thr=2.0, pt_seed=21, n_d=6),
    "pmvo_patch4": dict(V=22, H=128, W=96, seed=9, scale=1.7, rings=2, quantize=False, res=48, N=96, patch=4,
                        thr=0.05, vis_thr=0.5, pt_seed=23, n_d=8),
}


def scene_checksums(scene):
    return np.array([float(scene[k].double().sum()) for k in ("depth", "ori", "conf", "mask")], dtype=np.float64)


def ref_cameras(R, scene):
    cams = {}
    for c in scene["cams"]:
        cams[c["file"]] = R["Camera_utils"].Camera(c["ndc_prj"], np.linalg.inv(np.array(c["pose"])), c["file"])
    return cams


def pick_points(case):
    pts = synth.candidate_points(res=case["res"], seed=case["pt_seed"])
    rng = np.random.default_rng(case["pt_seed"])
    c = case.get("cluster", 1)
    sel = np.sort(rng.choice(len(pts), case["N"] // c, replace=False))
    out = pts[sel]
    if c > 1:       # c copies of every seed point, a few float32 ulps apart: same view ranking, different arithmetic
        out = np.repeat(out, c, axis=0) + rng.normal(0, 5e-8, size=(len(out) * c, 3))
    return out


def gen_pmvo(R, name, case):
    scene = synth.make_scene(case["V"], case["H"], case["W"], seed=case["seed"], scale=case["scale"],
                             rings=case["rings"], quantize=case["quantize"])
    cams = ref_cameras(R, scene)
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
    pm = R["PMVO"].PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                        patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
    pts64 = pick_points(case)
    out = dict(points=pts64)
    out["scene_checksums"] = scene_checksums(scene)
    if case.get("store_scene"):
        for k in ("depth", "ori", "conf", "mask"):
            out["scene_" + k] = scene[k].numpy()
    out["cam_pose"] = np.stack([c.pose.numpy() for c in pm.camera])
    out["cam_proj"] = np.stack([c.proj.numpy() for c in pm.camera])
    out["cam_rinv"] = np.stack([torch.linalg.inv(c.pose[:3, :3]).numpy() for c in pm.camera])

    # --- project_points on view 0 and view V//2 (PMVO.py:378-397)
    pts = torch.from_numpy(pts64).type(torch.float)
    for tag, vi in (("a", 0), ("b", case["V"] // 2)):
        uv, z, oob = pm.project_points(pts, pm.camera[vi], pm.image_size)
        out["proj_%s_view" % tag] = np.int32(vi)
        out["proj_%s_rc" % tag] = uv.numpy().astype(np.int32)
        out["proj_%s_z" % tag] = z.numpy()
        out["proj_%s_oob" % tag] = oob.numpy()

    # --- Compute_Visible_and_Ori (PMVO.py:346-376)
    pm.Compute_Visible_and_Ori(pts)
    out["visible"] = pm.visible.numpy()
    out["Ori"] = pm.Ori.numpy()
    out["Conf"] = pm.Conf.numpy()
    out["mask"] = pm.mask.numpy()
    nd = case["n_d"]
    out["Ori_patch_head"] = pm.Ori_patch[:, :nd].numpy()     # first n_d points only (size)
    out["Conf_patch_head"] = pm.Conf_patch[:, :nd].numpy()
    out["Ori_patch_sum"] = pm.Ori_patch.double().sum(dim=(2, 3)).numpy()   # [V,N] checksums for the rest
    out["Conf_patch_sum"] = pm.Conf_patch.double().sum(dim=2).numpy()

    # --- Find_max_conf_from_visible_view (PMVO.py:339-343)
    bidx, bval = pm.Find_max_conf_from_visible_view()
    out["base_idx"] = bidx.numpy().astype(np.int32)
    out["base_val"] = bval.numpy()

    # --- sample_next_3d_pos / compute_reproject_ori / compute_prj_loss for ranks 0 and 2
    for rank in (0, 2):
        samples, surface = pm.sample_next_3d_pos(pts, bidx[rank])
        assert torch.equal(surface, pts)  # PMVO.py:333-334 are no-ops
        D = pm.compute_reproject_ori(surface, samples)
        loss, idx, hc = pm.compute_prj_loss(D, pm.Ori, None)
        out["samples_r%d" % rank] = samples.numpy()
        out["D_head_r%d" % rank] = D[:, :nd].numpy()
        out["D_sum_r%d" % rank] = D.double().sum(dim=(2, 3)).numpy()
        out["loss_r%d" % rank] = loss.numpy()
        out["idx_r%d" % rank] = idx.numpy().astype(np.int32)
        out["hc_r%d" % rank] = hc.numpy()

    # --- forward (PMVO.py:39-78)
    sp, so, ml, hci = pm.forward(pts64)
    out["fwd_ori"] = so.numpy()
    out["fwd_loss"] = ml.numpy()
    out["fwd_hc"] = hci.numpy()

    # --- filter_points / compute_unvisible_points (PMVO.py:402-480)
    big = synth.candidate_points(res=case["res"], seed=case["pt_seed"] + 1)
    rng = np.random.default_rng(case["pt_seed"] + 1)
    big = big[np.sort(rng.choice(len(big), min(len(big), 1500), replace=False))]
    # push a third of them outward/inward so that every branch of the filters is exercised
    scale = np.ones(len(big))
    scale[::3] = 1.04
    scale[1::3] = 0.93
    big = big * scale[:, None]
    bigt = torch.from_numpy(big).type(torch.float)
    sidx, spts, fidx = pm.filter_points(bigt)
    out["filter_points_in"] = big
    out["filter_surface_index"] = sidx.numpy()
    out["filter_filter_index"] = fidx.numpy()
    out["unvisible_index"] = pm.compute_unvisible_points(bigt).numpy()

    # --- method refine (PMVO.py:81-144) with a toy bust / scalp (module globals of the reference)
    from scipy.spatial import KDTree

    rngb = np.random.default_rng(123)
    bust = rngb.normal(size=(500, 3))
    bust = bust / np.linalg.norm(bust, axis=1, keepdims=True) * 0.09
    scalp = bust[bust[:, 1] > 0.03] * (0.1 / 0.09)
    R["PMVO"].bust_tree = KDTree(data=bust)
    R["PMVO"].scalp_tree = KDTree(data=scalp)
    R["PMVO"].scalp_max = np.max(scalp, axis=0)
    out["toy_bust"] = bust
    out["toy_scalp"] = scalp
    ori_in = so.clone()
    bad = torch.isnan(ori_in).any(dim=1)
    ori_in[bad] = torch.tensor([0.0, -1.0, 0.0])
    rl = pm.refine(pts, ori_in)
    out["refine_ori_in"] = ori_in.numpy()
    out["refine_loss"] = rl.numpy()

    os.makedirs(OUT, exist_ok=True)
    meta = {k: v for k, v in case.items()}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=np.array(repr(meta)), **out)
    print(name, "written;", "nan losses:", int(np.isnan(out["fwd_loss"]).sum()), "/", case["N"])


def gen_offsets(R):
    ns = 90
    s1 = torch.arange(-0.005, -0.001, 0.004 / (ns / 4))
    s2 = torch.arange(-0.001, 0.001, 0.002 / (ns / 2))
    s3 = torch.arange(0.001, 0.005, 0.004 / (ns / 4))
    s = torch.cat([s1, s2, s3], 0)[:ns]
    np.save(os.path.join(OUT, "depth_offsets.npy"), s.numpy())
    print("depth_offsets written", s.shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    os.chdir("/tmp")
    torch.manual_seed(0)
    R = import_reference(gabor=False)
    if a.only in (None, "offsets"):
        gen_offsets(R)
    for name, case in PMVO_CASES.items():
        if a.only in (None, name):
            gen_pmvo(R, name, case)
    if a.only in (None, "consensus", "gabor", "e2e"):
        try:
            import gen_golden_more  # consensus / voxel-fit / gabor fixtures (added with those components)

            gen_golden_more.main(a.only)
        except ImportError:
            pass


if __name__ == "__main__":
    main()
