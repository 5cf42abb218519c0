"""tests/golden/pmvo_recompose.npz: the reference's own PMVO.forward on the SAME points in other batch compositions.

Why: on 1-2 % of the points of some fixtures (23 % of pmvo_views300) the reference's forward() and the oracle / HIP path
pick a different (rank, sample).  The cause lies inside the reference: Camera.reprojection multiplies a 3x3 matrix with
the [3, n] block of the points that share a base view (/root/reference/Utils/Camera_utils.py:81-109), and MKL takes
another kernel (different rounding) when a base view owns exactly ONE point of the batch.  Which points that happens to
depends on what else is in the batch -- so the reference's answer for a point is not a function of the point alone.
This script demonstrates it and pins the composition-independent answer:

  rev_*   forward(points[::-1])[::-1]           -- the same points in another order
  dup_*   forward(concat(points, points))[:N]   -- every base view owns >= 2 points at every rank: MKL's gemm kernel
                                                   everywhere, the form the oracle restates

The same doubled batch is recorded STEP BY STEP as well (round 4), for the step-level oracle tests
(tests/test_oracle_golden.py::test_sample_reproject_loss): for base-view ranks 0 and 2 of the doubled batch,

  dup_samples_r<k>   sample_next_3d_pos (PMVO.py:263-335)          first N rows  [N,90,3]
  dup_Dhead_r<k>     compute_reproject_ori (PMVO.py:222-241)       first n_d points [V,n_d,90,2]
  dup_Dsum_r<k>      its float64 checksums per (view, point)       [V,N]
  dup_loss_r<k>, dup_idx_r<k>, dup_hc_r<k>   compute_prj_loss (PMVO.py:151-220), first N rows

In the doubled batch every base view owns >= 2 points (MKL's gemm kernel, not gemv), and the first N points are not in the
trailing (2N*90 mod 64) columns that ATen's [V, N*S] sums add in another order -- so the oracle must equal these on EVERY row.

    python tools/gen_golden_recompose.py          (build container only: imports /root/reference)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)
sys.path.insert(2, os.path.join(ROOT, "tests"))

import gen_golden as G  # noqa: E402
from ref_import import import_reference  # noqa: E402
from monohair_amd import synth  # noqa: E402


def main():
    from conftest import golden_scene, load_golden

    os.chdir("/tmp")
    R = import_reference(gabor=False)
    out = {}
    for name, case in G.PMVO_CASES.items():
        meta, z = load_golden(name)
        scene = golden_scene(meta)           # (the stored scene where the fixture carries one)
        cams = G.ref_cameras(R, scene)
        depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
        pm = R["PMVO"].PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                            patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
        pts = z["points"]
        N = len(pts)

        def fwd(p):
            _, so, ml, hc = pm.forward(p)
            return so.numpy(), ml.numpy(), hc.numpy()

        o0, l0, h0 = fwd(pts)
        # the committed fixture is reproduced first (same container, same reference)
        assert np.array_equal(l0, z["fwd_loss"], equal_nan=True) and np.array_equal(o0, z["fwd_ori"], equal_nan=True)
        orv, lrv, hrv = (a[::-1].copy() for a in fwd(pts[::-1].copy()))
        od, ld, hd = fwd(np.concatenate([pts, pts], 0))
        # the two halves of the doubled batch agree with each other: this answer does not depend on the position
        assert np.array_equal(od[:N], od[N:], equal_nan=True) and np.array_equal(ld[:N], ld[N:], equal_nan=True)
        # the steps of the doubled batch (ranks 0 and 2, as tools/gen_golden.py records them for the original batch)
        import torch

        nd = case["n_d"]
        p2 = torch.from_numpy(np.concatenate([pts, pts], 0)).type(torch.float)
        pm.Compute_Visible_and_Ori(p2)
        bidx, bval = pm.Find_max_conf_from_visible_view()
        assert np.array_equal(bidx[:, :N].numpy(), z["base_idx"]) and np.array_equal(bidx[:, N:].numpy(), z["base_idx"])
        for rank in (0, 2):
            samples, surface = pm.sample_next_3d_pos(p2, bidx[rank])
            D = pm.compute_reproject_ori(surface, samples)
            loss, idx, hc = pm.compute_prj_loss(D, pm.Ori, None)
            assert torch.equal(samples[:N], samples[N:])
            out["%s__dup_samples_r%d" % (name, rank)] = samples[:N].numpy()
            out["%s__dup_Dhead_r%d" % (name, rank)] = D[:, :nd].numpy()
            out["%s__dup_Dsum_r%d" % (name, rank)] = D[:, :N].double().sum(dim=(2, 3)).numpy()
            out["%s__dup_loss_r%d" % (name, rank)] = loss[:N].numpy()
            out["%s__dup_idx_r%d" % (name, rank)] = idx[:N].numpy().astype(np.int32)
            out["%s__dup_hc_r%d" % (name, rank)] = hc[:N].numpy()
        for tag, (o, l, h) in (("rev", (orv, lrv, hrv)), ("dup", (od[:N], ld[:N], hd[:N]))):
            out["%s__%s_ori" % (name, tag)] = o
            out["%s__%s_loss" % (name, tag)] = l
            out["%s__%s_hc" % (name, tag)] = h
        same = lambda a, b: (a == b) | (np.isnan(a) & np.isnan(b))      # noqa: E731
        self_dis = ~(same(l0, ld[:N]) & np.all(same(o0, od[:N]), 1)) | ~(same(l0, lrv) & np.all(same(o0, orv), 1))
        print("%-16s N=%3d  reference(orig) != reference(recomposed) on %d rows" % (name, N, int(self_dis.sum())))
    np.savez_compressed(os.path.join(G.OUT, "pmvo_recompose.npz"), **out)
    print("pmvo_recompose.npz written")


if __name__ == "__main__":
    main()
