"""GPU parity for the rest of the path: filters, refine loss, consensus, voxel fit, Gabor bank and the
optimize -> refine -> .mat drivers, against the CPU oracle (exact) and the reference's goldens."""
import ast
import os
import types

import numpy as np
import pytest
import torch
from scipy.spatial import KDTree

import oracle
from conftest import GOLDEN, golden_records, golden_scene, load_golden, scene_views

pytestmark = pytest.mark.gpu
CASES = ["pmvo_small", "pmvo_mid", "pmvo_quant", "pmvo_views300", "pmvo_views300c", "pmvo_patch9", "pmvo_patch4"]
DEV = "cuda:0"


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def make_pmvo(meta, scene, rec):
    from monohair_amd.pmvo import PMVO

    return PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                            scene["mask"].to(DEV), device=DEV, patch_size=meta["patch"],
                            visible_threshold=meta["vis_thr"], conf_threshold=meta["thr"])


@pytest.fixture(scope="module", params=CASES)
def case(request):
    meta, z = load_golden(request.param)
    scene = golden_scene(meta)
    rec = golden_records(z)
    return meta, z, scene, scene_views(scene, rec), make_pmvo(meta, scene, rec)


def test_filters_vs_oracle_and_golden(case):
    meta, z, scene, views, pm = case
    pts = z["filter_points_in"]
    surf, spts, filt = pm.filter_points(torch.from_numpy(pts).to(DEV).float())
    unv = pm.compute_unvisible_points(torch.from_numpy(pts).to(DEV).float())
    o_surf, o_filt, o_unv, o_head = oracle.filter_votes(views, pts, meta["patch"], meta["thr"], meta["vis_thr"])
    assert np.array_equal(surf.cpu().numpy(), o_surf) and np.array_equal(filt.cpu().numpy(), o_filt)
    assert np.array_equal(unv.cpu().numpy(), o_unv)
    assert np.array_equal(surf.cpu().numpy(), z["filter_surface_index"])
    assert np.array_equal(filt.cpu().numpy(), z["filter_filter_index"])
    assert np.array_equal(unv.cpu().numpy(), z["unvisible_index"])
    assert torch.equal(spts.cpu(), torch.from_numpy(pts).float()[surf.cpu()])


def test_refine_method_vs_oracle_and_golden(case):
    meta, z, scene, views, pm = case
    pts = z["points"]
    scalp = z["toy_scalp"]
    pm.set_head(KDTree(data=z["toy_bust"]), KDTree(data=scalp), np.max(scalp, axis=0))
    loss = pm.refine(torch.from_numpy(pts).to(DEV).float(), torch.from_numpy(z["refine_ori_in"]).to(DEV))
    loss = loss.cpu().numpy()
    o_loss, _ = oracle.refine_loss(views, pts, z["refine_ori_in"], meta["patch"], meta["thr"])
    keep = loss != -1
    assert np.array_equal(loss[keep], o_loss[keep], equal_nan=True)          # exact vs oracle
    ref = z["refine_loss"]
    assert np.array_equal(loss == -1, ref == -1)                              # same filter decisions
    assert np.array_equal(loss, ref, equal_nan=True)      # the reference, every row (the trailing N mod 32 points included)


def test_consensus_vs_oracle_and_golden():
    """mh_medoid_dense == oracle == the reference (100 %): the mean runs in ATen's inner-dimension summation order
    (consensus.hip MhInnerSum); consensus_more.npz walks through every branch of that order, including groups of
    512+ members (second cascade level, the four-level kernel form)."""
    from monohair_amd.pmvo_utils import compute_points_similarity

    for name in ("consensus", "consensus_more"):
        z = load_npz(name)
        for k in sorted(f[:-3] for f in z.files if f.endswith("_in")):
            got, idx = compute_points_similarity(torch.from_numpy(z[k + "_in"]).to(DEV), return_index=True)
            o_out, o_idx = oracle.medoid_dense(z[k + "_in"])
            assert np.array_equal(idx.cpu().numpy(), o_idx), (name, k)
            assert np.array_equal(got.cpu().numpy(), o_out, equal_nan=True), (name, k)
            ref = z[k + "_out"]
            ok = np.all((got.cpu().numpy() == ref) | (np.isnan(ref) & np.isnan(got.cpu().numpy())), axis=1)
            assert ok.all(), (name, k, float(ok.mean()))


def test_segmented_medoid_mixes_small_and_large_groups():
    """one segmented call over groups of 1 .. 1300 members (both kernel forms of mh_medoid_segmented) == per-group oracle"""
    import ctypes

    from monohair_amd import _lib
    from monohair_amd.pmvo_utils import _ctx_for

    rng = np.random.default_rng(12)
    sizes = [1, 2, 5, 7, 8, 9, 31, 33, 100, 511, 512, 513, 640, 1300, 3, 64]
    seg = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.normal(size=(len(sizes), 3))
    ori = np.concatenate([b + 0.1 * rng.normal(size=(n, 3)) for b, n in zip(base, sizes)]).astype(np.float32)
    o = torch.from_numpy(ori).to(DEV)
    out = torch.empty((len(sizes), 3), device=DEV)
    idx = torch.empty((len(sizes),), dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().mh_medoid_segmented(_ctx_for(DEV), _lib.ptr(o), _lib.ptr(torch.from_numpy(seg).to(DEV)),
                                              len(sizes), max(sizes), _lib.ptr(out), _lib.ptr(idx), _lib.stream_ptr()))
    want, widx = oracle.medoid_segmented(ori, seg)
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(out.cpu().numpy(), want)


def test_voxel_fit_vs_oracle_and_golden():
    from monohair_amd.pmvo_utils import voxel_fit

    z = load_npz("e2e_small")
    meta = ast.literal_eval(str(z["meta"]))
    keep = np.where(z["ref_min_loss"] < meta["threshold"])[0]
    sel_o = np.concatenate([z["ref_select_o"][keep], z["ref_filter_unvisible_ori"]], 0)
    sel_p = np.concatenate([z["ref_select_p"][keep], z["ref_filter_unvisible"]], 0)
    res = voxel_fit(sel_p.copy(), sel_o.copy(), DEV)
    occ_o, ori_o = oracle.voxel_fit(sel_p.copy(), sel_o.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
    assert np.array_equal(res["occ"], occ_o)
    assert np.array_equal(res["ori_dense"], ori_o)
    ori_l, occ_l = oracle.mat_layout(res["occ"], res["ori_dense"])
    nz = np.argwhere(occ_l != 0).astype(np.int32)
    assert np.array_equal(nz, z["mat_occ_nz"])
    Z = occ_l.shape[2]
    got = np.stack([ori_l[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    assert np.all(got == z["mat_ori_at_nz"], axis=1).all()            # every voxel's orientation, bit for bit


def test_gabor_vs_oracle_and_golden():
    from monohair_amd.gabor import calOrientationGabor, gabor_bank

    z = load_npz("gabor")
    # same formula, same torch ops; the last bit of torch's CPU sin/cos/exp depends on the host CPU type, so
    # the reference's own kernels are installed for the bitwise comparisons below
    assert np.allclose(gabor_bank(), z["bank"], rtol=0, atol=2e-7)
    gab = calOrientationGabor(device=DEV, bank=z["bank"])
    assert gab.variant == "mfma2"                       # the shipped default is the first subject of every comparison
    for name in ("stripes0", "stripes30", "stripes90", "stripes135", "noise", "mixed"):
        img = z[name + "_img"]
        t = torch.from_numpy(img)[None, None].to(DEV)
        o_idx, o_conf, o_var = oracle.gabor_bank(z["bank"], img)
        ref_best, ref_conf, ref_two = z[name + "_best"], z[name + "_conf"], z[name + "_two"]
        for variant in ("mfma2", "valu"):
            gab.set_variant(variant)
            two, best, conf = gab(t, None, 1, threshold=0.0)
            idx, c2, var = gab.filter_index(t[0, 0])
            assert np.array_equal(idx.cpu().numpy(), o_idx), (name, variant)                 # exact vs oracle
            assert np.array_equal(var.cpu().numpy(), o_var), (name, variant)
            assert np.array_equal(c2.cpu().numpy(), o_conf), (name, variant)
            assert np.array_equal(best[0, 0].cpu().numpy(), ref_best), (name, variant)    # radians, bitwise, every pixel
            cf = conf[0, 0].cpu().numpy()
            assert np.allclose(cf, ref_conf, rtol=0, atol=1.2e-7) and (cf == ref_conf).mean() >= 0.998   # <= 1 ulp, rarely
            # ... and never across a boundary of the 8-bit code the reference writes to conf/<view>.png (what PMVO reads)
            code = lambda x: np.clip(x.astype(np.float32) * np.float32(255) + np.float32(0.5), 0, 255).astype(np.uint8)   # noqa: E731
            assert np.array_equal(code(cf), code(ref_conf)), (name, variant)
            assert np.array_equal(two[0].cpu().numpy(), ref_two)
            assert two.shape == (1, 2) + img.shape and best.shape == (1, 1) + img.shape
    gab.set_variant("mfma2")
    # the iterated form with a confidence threshold (forward(..., iter=2, threshold=0.3)) and the class's own
    # filter() / gabor_fn() with the reference's signatures
    t = torch.from_numpy(z["mixed_img"])[None, None].to(DEV)
    two, best, conf = gab(t, None, 2, threshold=0.3)
    agree = best[0, 0].cpu().numpy() == z["mixed_iter2_best"]
    # The second pass filters the CONFIDENCE of the first (GaborFilter.py:104-106), which equals the reference's to within one
    # float32 ulp on < 0.2 % of the pixels (above): a pixel of the second pass can differ only if such a pixel lies inside
    # its 17 x 17 window.  Asserted: every disagreeing pixel has one, and they are few.
    from scipy.ndimage import binary_dilation

    _, _, conf1 = gab(t, None, 1, threshold=0.0)
    seed = conf1[0, 0].cpu().numpy() != z["mixed_conf"]
    reach = binary_dilation(seed, structure=np.ones((17, 17), bool))
    assert not (~agree & ~reach).any(), int((~agree & ~reach).sum())
    assert agree.mean() >= 0.995, agree.mean()
    assert np.allclose(conf[0, 0].cpu().numpy()[agree], z["mixed_iter2_conf"][agree], rtol=0, atol=1e-6)
    assert np.allclose(two[0].cpu().numpy()[:, agree], z["mixed_iter2_two"][:, agree], rtol=0, atol=1e-7)
    k7 = gab.gabor_fn(17, 1, 1, torch.ones(1) * (np.pi * 7 / 180), 1.8, 2.4, 4)
    assert k7.shape == (1, 1, 17, 17) and np.allclose(k7[0, 0].cpu().numpy(), z["bank"][7], rtol=0, atol=2e-7)
    zero = torch.zeros_like(t)
    c1, v1, o1 = gab.filter(t, None, 0.0, zero, zero, zero, sigma_x=1.8, sigma_y=2.4, Lambda=4, kernel_size=17)
    _, b1, cf1 = gab(t, None, 1, threshold=0.0)
    assert torch.equal(o1, b1) and torch.equal(c1, cf1) and float(v1.max()) == 1.0


@pytest.mark.parametrize("variant", ["mfma2", "valu"])
def test_gabor_odd_sizes_and_border(variant):
    """ragged sizes (not multiples of the pixel tiles) incl. an image smaller than the kernel; both kernel variants
    (direct v_pk_fma form and the FP32-MFMA im2col contraction) are bit-identical to the oracle"""
    from monohair_amd.gabor import calOrientationGabor, gabor_bank

    gab = calOrientationGabor(device=DEV, variant=variant)
    rng = np.random.default_rng(3)
    for shape in ((9, 7), (17, 33), (50, 31), (64, 96), (75, 130)):
        img = rng.normal(size=shape).astype(np.float32)
        idx, conf, var = gab.filter_index(torch.from_numpy(img).to(DEV))
        o_idx, o_conf, o_var = oracle.gabor_bank(gabor_bank(), img)
        assert np.array_equal(idx.cpu().numpy(), o_idx) and np.array_equal(conf.cpu().numpy(), o_conf)


def _e2e_setup(tmp_path):
    z = load_npz("e2e_small")
    meta = ast.literal_eval(str(z["meta"]))
    scene = golden_scene(meta)
    pm = make_pmvo(meta, scene, golden_records(z))      # the reference's own camera tensors (see conftest)
    scalp = z["toy_scalp"]
    pm.set_head(KDTree(data=z["toy_bust"]), KDTree(data=scalp), np.max(scalp, axis=0))
    args = types.SimpleNamespace(device=DEV, output_path=str(tmp_path), save_root=str(tmp_path / "optimize"),
                                 save_path=str(tmp_path / "refine"),
                                 PMVO=types.SimpleNamespace(visible_threshold=meta["vis_thr"]),
                                 data=types.SimpleNamespace(root=str(tmp_path)))
    os.makedirs(args.save_path, exist_ok=True)
    return z, meta, pm, args


def test_drivers_end_to_end_vs_reference(tmp_path):
    """filter_negative_points -> optimize -> refine -> Ori3D/Occ3D.mat with the reference's file names and
    dtypes, against the reference's own run of the same pass (tests/golden/e2e_small.npz)."""
    import scipy.io

    from monohair_amd.pmvo import filter_negative_points, optimize, refine
    from monohair_amd.pmvo_utils import get_ground_truth_3D_occ, get_ground_truth_3D_ori

    z, meta, pm, args = _e2e_setup(tmp_path)
    cand = z["candidates"]
    surface_index, surface_points, filter_index = filter_negative_points(cand, pm, args)
    assert np.array_equal(surface_index, z["surface_index"]) and np.array_equal(filter_index, z["filter_index"])
    os.makedirs(args.save_root, exist_ok=True)
    np.save(os.path.join(args.save_root, "filter_unvisible.npy"), cand[filter_index])

    optimize(surface_points, pm, args)
    got = {k: np.load(os.path.join(args.save_root, k + ".npy")) for k in
           ("select_p", "select_o", "min_loss", "high_conf_index")}
    assert got["select_p"].dtype == np.float32 and got["select_o"].dtype == np.float32
    assert got["min_loss"].dtype == np.float32 and got["high_conf_index"].dtype == np.bool_
    assert np.array_equal(got["select_p"], z["opt_select_p"])
    same = (got["min_loss"] == z["opt_min_loss"]) | (np.isnan(got["min_loss"]) & np.isnan(z["opt_min_loss"]))
    same &= np.all((got["select_o"] == z["opt_select_o"]) | np.isnan(z["opt_select_o"]), axis=1)
    # the base-view ranking returns tied confidences in torch.topk's own order (74 % of these points have tied positive
    # values in their top 20 -- the synthetic confidences saturate at 1.0), so every selection is the reference's
    assert same.all(), same.mean()          # bit for bit on every candidate: the 1e-4 L-inf tolerance is met with 0

    # refine from the REFERENCE's optimize outputs, so the two refine stages see identical inputs
    fu = cand[filter_index]
    occ, ori = refine(z["opt_select_p"].copy(), z["opt_select_o"].copy(), z["opt_min_loss"].copy(), pm, fu, args,
                      infer_inner=False, threshold=meta["threshold"], genrate_ori_only=False)
    r = {k: np.load(os.path.join(str(tmp_path), "refine", k + ".npy")) for k in
         ("select_p", "select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori")}
    assert np.array_equal(r["select_p"], z["ref_select_p"])
    lm = (r["min_loss"] == z["ref_min_loss"]) | (np.isnan(r["min_loss"]) & np.isnan(z["ref_min_loss"]))
    om = np.all((r["select_o"] == z["ref_select_o"]) | (np.isnan(r["select_o"]) & np.isnan(z["ref_select_o"])), 1)
    # the medoid is bit-faithful to the reference (ATen's summation order), so the smoothing loop is too: every
    # orientation and every loss (the trailing N mod 32 points of a chunk, whose [V,N,1] sums ATen adds in its row_sum order,
    # included)
    assert om.all() and lm.all(), (lm.mean(), om.mean())
    assert np.array_equal(r["filter_unvisible"], z["ref_filter_unvisible"])
    fm = np.all(r["filter_unvisible_ori"] == z["ref_filter_unvisible_ori"], axis=1)
    assert fm.all(), fm.mean()

    Ori3 = scipy.io.loadmat(os.path.join(args.save_path, "Ori3D.mat"))["Ori"]
    Occ3 = scipy.io.loadmat(os.path.join(args.save_path, "Occ3D.mat"))["Occ"]
    assert Ori3.dtype == np.float64 and Occ3.dtype == np.float64
    assert Ori3.shape == tuple(z["mat_ori_shape"]) and Occ3.shape == tuple(z["mat_occ_shape"])
    nz = np.argwhere(Occ3 != 0).astype(np.int32)
    ref_nz = z["mat_occ_nz"]
    a = set(map(tuple, nz.tolist()))
    b = set(map(tuple, ref_nz.tolist()))
    assert a == b, (len(a), len(b), len(a ^ b))        # same occupied voxels
    Z = Occ3.shape[2]
    got_o = np.stack([Ori3[ref_nz[:, 0], ref_nz[:, 1], c * Z + ref_nz[:, 2]] for c in range(3)], 1)
    vm = np.all(got_o == z["mat_ori_at_nz"], axis=1)
    assert vm.all(), vm.mean()
    # the consumer's readers (HairGrow.py:41-55) see the documented shapes
    assert get_ground_truth_3D_occ(os.path.join(args.save_path, "Occ3D.mat")).shape == (192, 256, 256, 1)
    assert get_ground_truth_3D_ori(os.path.join(args.save_path, "Ori3D.mat")).shape == (192, 256, 256, 3)


def test_infer_inner_second_pass_vs_reference(tmp_path):
    """`PMVO.py --PMVO.infer_inner --PMVO.optimize=` (PMVO.py:874-880 -> refine(genrate_ori_only=True, infer_inner=True),
    :653-764): resumed from the exterior pass's checkpoint files, with a synthetic ours/raw.npy, against the reference's
    own run of that pass (tests/golden/e2e_inner.npz): coarse.npy / coarse_ori.npy (compute_unvisible_points + the sign
    flip), the occluded-shell orientations, and every voxel of full/Ori3D.mat / Occ3D.mat."""
    import scipy.io

    from monohair_amd.pmvo import refine

    z, meta, pm, args = _e2e_setup(tmp_path)
    inn = load_npz("e2e_inner")
    args.save_path = str(tmp_path / "full")
    os.makedirs(args.save_path)
    os.makedirs(tmp_path / "ours")
    os.makedirs(tmp_path / "refine", exist_ok=True)
    for k in ("select_p", "select_o", "min_loss"):
        np.save(tmp_path / "refine" / (k + ".npy"), z["ref_" + k])
    np.save(tmp_path / "ours" / "raw.npy", inn["raw"])
    refine(z["opt_select_p"].copy(), z["opt_select_o"].copy(), z["opt_min_loss"].copy(), pm, inn["filter_unvisible_in"],
           args, infer_inner=True, threshold=meta["threshold"], genrate_ori_only=True, return_dense=False)
    unv = pm.compute_unvisible_points(torch.from_numpy(inn["raw"][:, :3].astype(np.float32)).to(DEV)).cpu().numpy()
    assert np.array_equal(unv, inn["unvisible_index"])
    assert np.array_equal(np.load(tmp_path / "full" / "coarse.npy"), inn["coarse"])
    assert np.array_equal(np.load(tmp_path / "full" / "coarse_ori.npy"), inn["coarse_ori"])
    assert np.array_equal(np.load(tmp_path / "refine" / "filter_unvisible.npy"), inn["ref_filter_unvisible"])
    assert np.array_equal(np.load(tmp_path / "refine" / "filter_unvisible_ori.npy"), inn["ref_filter_unvisible_ori"])
    Ori3 = scipy.io.loadmat(tmp_path / "full" / "Ori3D.mat")["Ori"]
    Occ3 = scipy.io.loadmat(tmp_path / "full" / "Occ3D.mat")["Occ"]
    assert Ori3.shape == tuple(inn["mat_ori_shape"]) and Occ3.shape == tuple(inn["mat_occ_shape"])
    nz = np.argwhere(Occ3 != 0).astype(np.int32)
    assert np.array_equal(nz, inn["mat_occ_nz"])
    Z = Occ3.shape[2]
    got = np.stack([Ori3[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    assert np.array_equal(got, inn["mat_ori_at_nz"])


def test_gabor_to_pmvo_device_handoff_equals_file_roundtrip(tmp_path):
    """orientation_maps_device (DoG + Gabor bank + 8-bit emulation, all on the GPU) gives exactly the maps that the
    reference-style file pipeline gives: calculate_orientation -> best_ori/conf PNGs -> Load_Ori_And_Conf."""
    from PIL import Image

    from monohair_amd.gabor import (batch_generate, difference_of_gaussians, difference_of_gaussians_device,
                                    orientation_maps_device)
    from monohair_amd.pmvo_utils import Load_Ori_And_Conf

    rng = np.random.default_rng(5)
    H, W, V = 96, 72, 3
    os.makedirs(tmp_path / "capture_images")
    os.makedirs(tmp_path / "hair_mask")
    imgs = []
    for v in range(V):
        r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        th = np.deg2rad(25 + 40 * v)
        im = (127 + 70 * np.cos(2 * np.pi * (r * np.cos(th) + c * np.sin(th)) / 4.0) + rng.normal(0, 6, (H, W)))
        im = im.clip(0, 255).astype(np.uint8)
        imgs.append(im)
        Image.fromarray(im).save(tmp_path / "capture_images" / ("v%d.png" % v))
        Image.fromarray(np.full((H, W), 255, np.uint8)).save(tmp_path / "hair_mask" / ("v%d.png" % v))
    # device DoG == scipy DoG (float64 torch ops in scipy's accumulation order)
    a = difference_of_gaussians(imgs[0], 0.4, 10)
    b = difference_of_gaussians_device(imgs[0], 0.4, 10, DEV).cpu().numpy()
    assert np.array_equal(a, b)
    # ... and both == the real scikit-image (tests/golden/dog.npz; 1e-15: numpy's exp in the Gaussian weights differs in the
    # last bit between numpy versions)
    zd = np.load(os.path.join(GOLDEN, "dog.npz"))
    for k in ("stripes", "noise", "ramp", "small", "codes"):
        d = difference_of_gaussians_device(zd["in_" + k], 0.4, 10, DEV).cpu().numpy()
        assert np.array_equal(d, difference_of_gaussians(zd["in_" + k], 0.4, 10))
        assert np.abs(d - zd["dog_" + k]).max() <= 1e-15
    ori, conf = orientation_maps_device(imgs, device=DEV)
    batch_generate(str(tmp_path), "capture_images")
    cam = {"v%d" % v: None for v in range(V)}
    Ori, Conf = Load_Ori_And_Conf(cam, str(tmp_path / "best_ori"), str(tmp_path / "conf"))
    for v in range(V):
        assert np.array_equal(ori[v].cpu().numpy(), Ori["v%d" % v].astype(np.float32))
        assert np.array_equal(conf[v].cpu().numpy(), Conf["v%d" % v].astype(np.float32))
    assert ori.shape == (V, H, W, 2) and float(conf.max()) == 1.0


@pytest.mark.parametrize("patch", [1, 3, 7, 11])
def test_refine_chunk_kernels_equal_the_unfused_path(patch):
    """mh_refine_loss_maps == mh_project_gather + mh_refine_loss, mh_medoid_indexed == the medoid of the gathered
    rows, mh_refine_combine == the tensor ops of the smoothing loop (PMVO.py:91-92, 631-642) -- bit for bit."""
    import ctypes

    from monohair_amd import _lib, synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO
    from monohair_amd.pmvo_utils import compute_points_similarity

    V, H, W, N, K = 40, 150, 110, 700, 37
    scene = synth.make_scene(V, H, W, seed=6, quantize=True, rings=2)
    pm = PMVO.from_planes(camera_records(cameras_from_list(scene["cams"])), scene["depth"].to(DEV), scene["ori"].to(DEV),
                          scene["conf"].to(DEV), scene["mask"].to(DEV), device=DEV, patch_size=patch, conf_threshold=0.15)
    g = torch.Generator().manual_seed(patch)
    pts = torch.from_numpy(synth.candidate_points(res=32, seed=patch)[:N]).float().to(DEV)
    ori_all = torch.randn((N, 3), generator=g).to(DEV)
    index = torch.randint(0, N, (N, K), generator=g, dtype=torch.int32).to(DEV)
    L, st = pm._L, _lib.stream_ptr()
    center = torch.empty((N, 3), device=DEV)
    _lib.check(L.mh_medoid_indexed(pm._ctx, _lib.ptr(ori_all), _lib.ptr(index), N, K, _lib.ptr(center), None, st))
    assert torch.equal(center, compute_points_similarity(ori_all[index.long()]))
    loss_f = torch.empty((N,), device=DEV)
    hc_f = torch.empty((N,), dtype=torch.uint8, device=DEV)
    _lib.check(L.mh_refine_loss_maps(pm._ctx, _lib.ptr(pts), _lib.ptr(center), 0.005, 4.0, N, patch, 0.15,
                                     _lib.ptr(loss_f), _lib.ptr(hc_f), 0, 0, 0, st))
    pm.Compute_Visible_and_Ori(pts)
    loss_u, hc_u = pm.prj_loss_of(pm._points, center)
    same = (loss_f == loss_u) | (torch.isnan(loss_f) & torch.isnan(loss_u))
    assert bool(same.all()) and torch.equal(hc_f.bool(), hc_u) and bool(torch.isfinite(loss_u).any())
    head = (torch.rand((N,), generator=g) < 0.3).to(torch.uint8).to(DEV)
    head_top = (torch.rand((N,), generator=g) < 0.5).to(torch.uint8).to(DEV)
    ori_new, loss_out = ori_all.clone(), torch.empty((N,), device=DEV)
    _lib.check(L.mh_refine_combine(pm._ctx, _lib.ptr(center), _lib.ptr(loss_u), _lib.ptr(head), _lib.ptr(head_top), 0.95,
                                   _lib.ptr(ori_new), _lib.ptr(loss_out), N, st))
    upd = torch.where(head.bool() & ~head_top.bool(), torch.full_like(loss_u, -1.0), loss_u)
    want_loss = torch.where(upd == -1, torch.full_like(upd, 0.5), upd)
    want_ori = ori_all.clone()
    pm.replace_dissimilar(center, want_ori, 0.95)
    assert torch.equal(ori_new, want_ori)
    assert bool(((loss_out == want_loss) | (torch.isnan(loss_out) & torch.isnan(want_loss))).all())
