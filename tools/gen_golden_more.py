"""More golden vectors from the imported reference (see gen_golden.py): orientation consensus, the Gabor bank,
and one end-to-end exterior pass (filter_negative_points -> optimize -> refine -> Ori3D/Occ3D .mat arrays).

    python tools/gen_golden.py --only consensus|gabor|e2e        (dispatches here)
"""
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)

from ref_import import import_reference  # noqa: E402
from monohair_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_consensus(R):
    cps = R["PMVO_utils"].compute_points_similarity
    rng = np.random.default_rng(42)
    out = {}
    # (a) random directions, K = 100 (the KNN use, PMVO.py:626)
    a = rng.normal(size=(50, 100, 3)).astype(np.float32)
    out["a_in"] = a
    out["a_out"] = cps(torch.from_numpy(a)).numpy()
    # (b) clustered directions with sign flips (what neighbourhoods of hair look like)
    base = rng.normal(size=(40, 1, 3))
    b = base + 0.15 * rng.normal(size=(40, 100, 3))
    b *= rng.choice([-1.0, 1.0], size=(40, 100, 1))
    b = b.astype(np.float32)
    out["b_in"] = b
    out["b_out"] = cps(torch.from_numpy(b)).numpy()
    # (c) degenerate: duplicates (exact ties -> first index), zero vectors, K = 1, 2, 17
    c = rng.normal(size=(6, 17, 3)).astype(np.float32)
    c[0, 5] = c[0, 2]
    c[0, 9] = -c[0, 2]
    c[1, :] = c[1, 0]
    c[2, 3] = 0
    c[3, :, :] = 0
    out["c_in"] = c
    out["c_out"] = cps(torch.from_numpy(c)).numpy()
    for K in (1, 2, 3):
        d = rng.normal(size=(8, K, 3)).astype(np.float32)
        out["d%d_in" % K] = d
        out["d%d_out" % K] = cps(torch.from_numpy(d)).numpy()
    np.savez_compressed(os.path.join(OUT, "consensus.npz"), **out)
    print("consensus written")


def stripes(H, W, theta_deg, period=4.0):
    """stripes whose oscillation axis is (row, col) = (cos th, sin th)"""
    th = np.deg2rad(theta_deg)
    r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return (0.5 + 0.25 * np.cos(2 * np.pi * (r * np.cos(th) + c * np.sin(th)) / period)).astype(np.float32)


def gen_gabor(R):
    G = R["GaborFilter"]
    gab = G.calOrientationGabor()
    out = {}
    bank = []
    for k in range(180):
        theta = torch.ones(1) * (np.pi * k / 180)
        bank.append(gab.gabor_fn(17, 1, 1, theta, 1.8, 2.4, 4)[0, 0].numpy())
    out["bank"] = np.stack(bank).astype(np.float32)
    rng = np.random.default_rng(7)
    imgs = {"stripes0": stripes(96, 96, 0), "stripes30": stripes(96, 96, 30), "stripes90": stripes(96, 96, 90),
            "stripes135": stripes(96, 96, 135), "noise": rng.normal(size=(64, 48)).astype(np.float32),
            "mixed": (stripes(80, 56, 70) + 0.05 * rng.normal(size=(80, 56))).astype(np.float32)}
    for name, im in imgs.items():
        t = torch.from_numpy(im)[None, None]
        two, best, conf = gab(t, torch.ones_like(t), 1, threshold=0.0)
        out[name + "_img"] = im
        out[name + "_best"] = best[0, 0].numpy()
        out[name + "_conf"] = conf[0, 0].numpy()
        out[name + "_two"] = two[0].numpy()
    # the iterated form (forward re-filters its own confidence map, GaborFilter.py:104-106) and a confidence threshold
    t = torch.from_numpy(imgs["mixed"])[None, None]
    two, best, conf = gab(t, torch.ones_like(t), 2, threshold=0.3)
    out["mixed_iter2_best"] = best[0, 0].numpy()
    out["mixed_iter2_conf"] = conf[0, 0].numpy()
    out["mixed_iter2_two"] = two[0].numpy()
    np.savez_compressed(os.path.join(OUT, "gabor.npz"), **out)
    print("gabor written")


E2E = dict(V=24, H=240, W=136, seed=0, scale=1.7, rings=1, quantize=False, res=32, patch=3, thr=0.15, vis_thr=1.0,
           threshold=0.05, pt_seed=4)


def gen_e2e(R):
    """The reference's own drivers end to end (PMVO.py:535-764) on the synthetic sphere."""
    from scipy.spatial import KDTree
    import scipy.io

    case = E2E
    ref = R["PMVO"]
    scene = synth.make_scene(case["V"], case["H"], case["W"], seed=case["seed"], scale=case["scale"],
                             rings=case["rings"], quantize=case["quantize"])
    cams = {}
    for c in scene["cams"]:
        cams[c["file"]] = R["Camera_utils"].Camera(c["ndc_prj"], np.linalg.inv(np.array(c["pose"])), c["file"])
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
    pm = ref.PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                  patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
    rngb = np.random.default_rng(123)
    bust = rngb.normal(size=(500, 3))
    bust = bust / np.linalg.norm(bust, axis=1, keepdims=True) * 0.09
    scalp = bust[bust[:, 1] > 0.03] * (0.1 / 0.09)
    ref.bust_tree = KDTree(data=bust)
    ref.scalp_tree = KDTree(data=scalp)
    ref.scalp_max = np.max(scalp, axis=0)
    ref.device = "cpu"
    tmp = tempfile.mkdtemp(prefix="mh_e2e_")
    args = types.SimpleNamespace()
    args.device = "cpu"
    args.output_path = tmp
    args.save_root = os.path.join(tmp, "optimize")
    args.save_path = os.path.join(tmp, "refine")
    os.makedirs(args.save_path, exist_ok=True)
    args.PMVO = types.SimpleNamespace(visible_threshold=case["vis_thr"])
    args.data = types.SimpleNamespace(root=tmp)
    points = synth.candidate_points(res=case["res"], seed=case["pt_seed"])
    out = dict(candidates=points.copy(), toy_bust=bust, toy_scalp=scalp)
    out["cam_pose"] = np.stack([c.pose.numpy() for c in pm.camera])
    out["cam_proj"] = np.stack([c.proj.numpy() for c in pm.camera])
    out["cam_rinv"] = np.stack([torch.linalg.inv(c.pose[:3, :3]).numpy() for c in pm.camera])
    surface_index, surface_points, filter_index = ref.filter_negative_points(points, pm, args)
    out["surface_index"] = surface_index
    out["filter_index"] = filter_index
    raw = points.copy()
    os.makedirs(args.save_root, exist_ok=True)
    np.save(os.path.join(args.save_root, "filter_unvisible.npy"), raw[filter_index])
    ref.Num_points = surface_points.shape[0]
    ref.optimize(surface_points, pm, args)
    for k in ("select_p", "select_o", "min_loss", "high_conf_index"):
        out["opt_" + k] = np.load(os.path.join(args.save_root, k + ".npy"))
    sp, so, ml = out["opt_select_p"].copy(), out["opt_select_o"].copy(), out["opt_min_loss"].copy()
    fu = np.load(os.path.join(args.save_root, "filter_unvisible.npy"))
    ref.refine(sp, so, ml, pm, fu, args, infer_inner=False, threshold=case["threshold"], genrate_ori_only=False)
    for k in ("select_p", "select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori"):
        out["ref_" + k] = np.load(os.path.join(tmp, "refine", k + ".npy"))
    Ori3 = scipy.io.loadmat(os.path.join(tmp, "refine", "Ori3D.mat"))["Ori"]
    Occ3 = scipy.io.loadmat(os.path.join(tmp, "refine", "Occ3D.mat"))["Occ"]
    out["mat_ori_shape"] = np.array(Ori3.shape)
    out["mat_occ_shape"] = np.array(Occ3.shape)
    nz = np.argwhere(Occ3 != 0)
    out["mat_occ_nz"] = nz.astype(np.int32)                      # [M,3] indices into Occ[Y,X,Z]
    Z = Occ3.shape[2]
    out["mat_ori_at_nz"] = np.stack([Ori3[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    out["mat_ori_nnz"] = np.array([np.count_nonzero(Ori3)])
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "e2e_small.npz"), meta=np.array(repr(case)), **out)
    print("e2e written: %d candidates, %d surface, %d shell, %d voxels" %
          (len(points), int(surface_index.sum()), int(filter_index.sum()), len(nz)))


def hair_volume(G=(40, 40, 36), R=11.0, seed=0):
    """A small synthetic fitted volume in the reference's array layout: occ [X,Y,Z], ori [X,Y,Z,3] (unit meridian
    tangents + noise on a spherical shell, as refine() would write them before the .mat layout transform)."""
    rng = np.random.default_rng(seed)
    x, y, z = np.meshgrid(*[np.arange(g) for g in G], indexing="ij")
    c = np.array([G[0] / 2, G[1] / 2, G[2] / 2])
    p = np.stack([x - c[0], y - c[1], z - c[2]], -1).astype(np.float64)
    r = np.linalg.norm(p, axis=-1)
    shell = np.abs(r - R) <= 1.0
    n = p / np.maximum(r[..., None], 1e-9)
    t = -np.array([0, 1.0, 0]) + n[..., 1:2] * n
    t = t / np.maximum(np.linalg.norm(t, axis=-1, keepdims=True), 1e-9)
    t = t + 0.12 * rng.normal(size=t.shape)
    t = t / np.linalg.norm(t, axis=-1, keepdims=True)
    occ = shell.astype(np.float64)
    ori = np.where(shell[..., None], t, 0.0)
    drop = rng.random(G) < 0.03          # a few occupied voxels without orientation / holes in the occupancy
    ori[drop] = 0
    occ[rng.random(G) < 0.01] = 0
    return occ, ori


def gen_hairgrow(R_):
    """HairGrowing.GenerateGuideStrandFromScalp / randomlyGenerateSegments (HairGrow.py:59-299) run by the reference."""
    import scipy.io
    import HairGrow

    G = (40, 40, 36)
    occ, ori = hair_volume(G)
    tmp = tempfile.mkdtemp(prefix="mh_hg_")
    o = ori.transpose((0, 1, 3, 2)).reshape(G[0], G[1], G[2] * 3).transpose((1, 0, 2))
    scipy.io.savemat(os.path.join(tmp, "Ori3D.mat"), {"Ori": o})
    scipy.io.savemat(os.path.join(tmp, "Occ3D.mat"), {"Occ": occ.transpose((1, 0, 2))})
    rng = np.random.default_rng(3)
    nrm = rng.normal(size=(400, 3))
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm[:, 1] = -np.abs(nrm[:, 1]) * 0.5                       # mostly pointing sideways / "up" in voxel space
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    # reader layout is [Z,Y,X]: positions are (x,y,z) voxel coordinates of THAT array
    centre = np.array([G[0] / 2, G[1] / 2, G[2] / 2])
    pts = (centre + nrm * 5.0 + rng.normal(0, 0.3, size=(400, 3))).astype(np.float32)
    nrm = nrm.astype(np.float32)
    out = dict(occ=occ.astype(np.float32), ori=ori.astype(np.float32), scalp_points=pts, scalp_normals=nrm,
               thr=np.float32(0.8))
    for name in ("guide", "random"):
        solver = HairGrow.HairGrowing(os.path.join(tmp, "Occ3D.mat"), os.path.join(tmp, "Ori3D.mat"), device="cpu")
        torch.manual_seed(77)
        if name == "guide":
            strands, num_root = solver.GenerateGuideStrandFromScalp(torch.from_numpy(pts.copy()),
                                                                    torch.from_numpy(nrm.copy()), None, 0.8)
            out["guide_num_root"] = np.int32(num_root)
        else:
            strands = solver.randomlyGenerateSegments(0.8)
        out[name + "_len"] = np.array([s.shape[0] for s in strands], np.int32)
        out[name + "_pts"] = torch.cat(strands, 0).numpy()
        print(name, "strands:", len(strands), "points:", out[name + "_pts"].shape[0])
    out["vol_occ_zyx"] = solver.occ[0].numpy()
    shutil.rmtree(tmp)
    np.savez_compressed(os.path.join(OUT, "hairgrow.npz"), **out)
    print("hairgrow written")


def main(only=None):
    os.makedirs(OUT, exist_ok=True)
    cwd = os.getcwd()
    os.chdir("/tmp")
    if only in (None, "consensus", "e2e", "hairgrow"):
        R = import_reference(gabor=False)
        if only in (None, "hairgrow"):
            gen_hairgrow(R)
        if only in (None, "consensus"):
            gen_consensus(R)
        if only in (None, "e2e"):
            gen_e2e(R)
    if only in (None, "gabor"):
        R = import_reference(gabor=True)
        gen_gabor(R)
    os.chdir(cwd)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
