"""Golden vectors for the CHUNKED drivers of the reference (run here, from /root/reference, CPU torch):

    python tools/gen_golden_multichunk.py          -> tests/golden/e2e_multichunk.npz

The reference walks its points in chunks of 5000 (`optimize`, /root/reference/PMVO.py:565-595; `refine`,
:602-650).  `optimize`'s chunks are independent; `refine`'s are NOT: `Neighbor_ori = ori[index]` (:612) reads the
array that earlier chunks already wrote back (:640), so chunk k+1 sees chunk k's replaced orientations.
tests/golden/e2e_small.npz is one chunk (3153 points) and cannot show that.  This fixture is the same synthetic
scene (V=24, 240x136, patch 3) with the res-64 candidate grid: >= 11 000 surface points = 3 chunks with a ragged
last one, plus a second `refine` run on exactly 10 000 of those points, which records what
`step = N // 5000 + 1` (:603) does with its empty trailing chunk.

The reference's forward() is batch-dependent at the last bit (MKL picks its gemm kernel in Camera.reprojection by the
number of points a base view owns; tools/gen_golden_recompose.py), so optimize is ALSO recorded in a second
batch composition: pieces of 125 points, each doubled (`forward(cat[sub, sub])`, first half kept; see
_optimize_recomposed for why this composition) -> `optrec_*`.  The parity statement
(tests/conftest.py::check_rows_against_recomposed) is: equal to the doubled-batch answer on every row, and every row that
differs from the four-chunk run is a row on which the reference disagrees with itself.

    python tools/gen_golden_multichunk.py --extend   recomputes only the optrec_* arrays and keeps the rest of an
                                                      existing file (same bytes as a full run; saves the other 6 minutes)

Stored: the candidates, the reference's filter_negative_points masks, optimize's four files, refine's five files
for both runs (or the exception text, if the reference raises), and the occupied voxels of Ori3D/Occ3D.mat.
"""
import os
import shutil
import sys
import tempfile
import time
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

CASE = dict(V=24, H=240, W=136, seed=0, scale=1.7, rings=1, quantize=False, res=64, patch=3, thr=0.15, vis_thr=1.0,
            threshold=0.05, pt_seed=4, exact=10000)


def _mats(tmp, out, prefix):
    import scipy.io

    Ori3 = scipy.io.loadmat(os.path.join(tmp, "refine", "Ori3D.mat"))["Ori"]
    Occ3 = scipy.io.loadmat(os.path.join(tmp, "refine", "Occ3D.mat"))["Occ"]
    nz = np.argwhere(Occ3 != 0)
    Z = Occ3.shape[2]
    out[prefix + "mat_occ_nz"] = nz.astype(np.int32)
    out[prefix + "mat_ori_at_nz"] = np.stack([Ori3[nz[:, 0], nz[:, 1], c * Z + nz[:, 2]] for c in range(3)], 1)
    return len(nz)


def _optimize_recomposed(pm, out, piece=125):
    """optimize's points (PMVO.py:565-579) through forward() in ANOTHER batch composition: pieces of 125 points, each
    doubled (`forward(cat[sub, sub])`, first half kept), so that every (rank, base view) group of a call holds between 2 and
    250 points.  Why this one: Camera.reprojection's matmul (Camera_utils.py:103) is [3,3] x [3, 90*n] for the n points of a
    group, and MKL's sgemm rounds differently by size -- gemv for n = 1, (a0*b0 + a2*b2) + a1*b1 up to ~28k columns
    (n <= 316), a k-ordered fma chain above (oracle/pmvo_oracle.c, cam_unproject).  With 5000-point chunks and 24 views some
    groups exceed 316 points, so the reference's four-chunk files mix the two roundings; this composition has the mid-size
    form everywhere -- the form the oracle and the HIP kernels pin."""
    pts = out["opt_select_p"].astype(np.float64)
    N = len(pts)
    o, l, h = [], [], []
    t0 = time.time()
    for a in range(0, N, piece):
        sub = pts[a:a + piece]
        n = len(sub)
        _, so, sl, sh = pm.forward(np.concatenate([sub, sub], 0))
        assert torch.equal(so[:n].isnan(), so[n:].isnan())
        o.append(so[:n]), l.append(sl[:n]), h.append(sh[:n])
    out["optrec_select_o"] = torch.cat(o, 0).numpy()
    out["optrec_min_loss"] = torch.cat(l, 0).numpy()
    out["optrec_high_conf_index"] = torch.cat(h, 0).numpy()
    print("optimize, recomposed (doubled %d-point pieces) %.1f s" % (piece, time.time() - t0))


def main():
    from scipy.spatial import KDTree

    torch.manual_seed(0)
    R = import_reference()
    ref = R["PMVO"]
    case = CASE
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

    def fresh_args():
        tmp = tempfile.mkdtemp(prefix="mh_e2e_mc_")
        a = types.SimpleNamespace(device="cpu", output_path=tmp, save_root=os.path.join(tmp, "optimize"),
                                  save_path=os.path.join(tmp, "refine"),
                                  PMVO=types.SimpleNamespace(visible_threshold=case["vis_thr"]),
                                  data=types.SimpleNamespace(root=tmp))
        os.makedirs(a.save_path, exist_ok=True)
        os.makedirs(a.save_root, exist_ok=True)
        return tmp, a

    extend = "--extend" in sys.argv
    if extend:
        old = np.load(os.path.join(OUT, "e2e_multichunk.npz"))
        assert str(old["meta"]) == repr(case)
        out = {k: old[k] for k in old.files if k != "meta" and not k.startswith(("optdup_", "optrec_"))}
        _optimize_recomposed(pm, out)
        np.savez_compressed(os.path.join(OUT, "e2e_multichunk.npz"), meta=np.array(repr(case)), **out)
        print("e2e_multichunk extended")
        return
    tmp, args = fresh_args()
    points = synth.candidate_points(res=case["res"], seed=case["pt_seed"])
    out = dict(candidates=points.copy(), toy_bust=bust, toy_scalp=scalp)
    out["cam_pose"] = np.stack([c.pose.numpy() for c in pm.camera])
    out["cam_proj"] = np.stack([c.proj.numpy() for c in pm.camera])
    out["cam_rinv"] = np.stack([torch.linalg.inv(c.pose[:3, :3]).numpy() for c in pm.camera])
    t0 = time.time()
    surface_index, surface_points, filter_index = ref.filter_negative_points(points, pm, args)
    out["surface_index"] = surface_index
    out["filter_index"] = filter_index
    print("filter_negative_points %.1f s: %d surface, %d shell" % (time.time() - t0, surface_index.sum(),
                                                                    filter_index.sum()))
    assert surface_points.shape[0] >= 11000 and surface_points.shape[0] % 5000 != 0
    fu = points[filter_index]
    ref.Num_points = surface_points.shape[0]
    t0 = time.time()
    ref.optimize(surface_points, pm, args)
    print("optimize %.1f s" % (time.time() - t0))
    for k in ("select_p", "select_o", "min_loss", "high_conf_index"):
        out["opt_" + k] = np.load(os.path.join(args.save_root, k + ".npy"))

    # refine, 3 chunks (ragged last)
    sp, so, ml = out["opt_select_p"].copy(), out["opt_select_o"].copy(), out["opt_min_loss"].copy()
    t0 = time.time()
    ref.refine(sp, so, ml, pm, fu.copy(), args, infer_inner=False, threshold=case["threshold"], genrate_ori_only=False)
    print("refine (%d points) %.1f s" % (len(sp), time.time() - t0))
    for k in ("select_p", "select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori"):
        out["ref_" + k] = np.load(os.path.join(tmp, "refine", k + ".npy"))
    nv = _mats(tmp, out, "")
    shutil.rmtree(tmp)

    # refine on exactly 10 000 points: step = 10000 // 5000 + 1 = 3, the third chunk is empty (PMVO.py:603-608)
    n = case["exact"]
    tmp, args = fresh_args()
    sp, so, ml = out["opt_select_p"][:n].copy(), out["opt_select_o"][:n].copy(), out["opt_min_loss"][:n].copy()
    fu2 = fu[:3000].copy()
    try:
        ref.refine(sp, so, ml, pm, fu2, args, infer_inner=False, threshold=case["threshold"], genrate_ori_only=False)
        out["exact_raised"] = np.array("")
        for k in ("select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori"):
            out["exact_" + k] = np.load(os.path.join(tmp, "refine", k + ".npy"))
        _mats(tmp, out, "exact_")
        print("refine on exactly %d points: ran through" % n)
    except Exception as e:       # recorded, and mirrored by the product path
        out["exact_raised"] = np.array("%s: %s" % (type(e).__name__, e))
        print("refine on exactly %d points RAISED %s: %s" % (n, type(e).__name__, e))
    shutil.rmtree(tmp)

    # optimize on exactly 10 000 points: Num_points // 5000 + 1 = 3 chunks, the third is forward() of zero points (:567-571)
    tmp, args = fresh_args()
    ref.Num_points = n
    try:
        ref.optimize(out["opt_select_p"][:n].astype(np.float64), pm, args)
        got = {k: np.load(os.path.join(args.save_root, k + ".npy")) for k in ("select_o", "min_loss", "high_conf_index")}
        out["exact_opt_raised"] = np.array("")
        out["exact_opt_equal_prefix"] = np.array(all(
            np.array_equal(got[k], out["opt_" + k][:n], equal_nan=(k != "high_conf_index")) for k in got))
        print("optimize on exactly %d points ran through; equal to the 3-chunk prefix: %s"
              % (n, bool(out["exact_opt_equal_prefix"])))
    except Exception as e:
        out["exact_opt_raised"] = np.array("%s: %s" % (type(e).__name__, e))
        print("optimize on exactly %d points RAISED %s: %s" % (n, type(e).__name__, e))
    shutil.rmtree(tmp)

    _optimize_recomposed(pm, out)
    np.savez_compressed(os.path.join(OUT, "e2e_multichunk.npz"), meta=np.array(repr(case)), **out)
    print("e2e_multichunk written: %d candidates, %d surface, %d shell, %d voxels" %
          (len(points), int(surface_index.sum()), int(filter_index.sum()), nv))


if __name__ == "__main__":
    main()
