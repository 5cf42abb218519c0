"""tests/golden/pmvo_single.npz: the reference on BATCHES OF ONE POINT (round 5).

A batch of one point makes every sgemm of Camera.projection a [4,4] x [4,1] product (/root/reference/Utils/Camera_utils.py:50-53),
which MKL rounds in its own way (tools/probe_mkl_forms.py) -- in Compute_Visible_and_Ori, in compute_reproject_ori's projection
of the point, in the votes of filter_points.  It happens in production: the last chunk of optimize / refine holds
`N mod 5000` points (/root/reference/PMVO.py:572-574, 604-606).  This fixture records, for points of two committed scenes each
handed to the reference ALONE: forward (PMVO.py:39-78), the method refine (:81-93) and the votes (:402-480).

    python tools/gen_golden_single.py          (build container only: imports /root/reference)
"""
import os
import sys

import numpy as np
import torch

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
    from scipy.spatial import KDTree

    os.chdir("/tmp")
    R = import_reference(gabor=False)
    out = {}
    for name in ("pmvo_small", "pmvo_quant"):
        case = G.PMVO_CASES[name]
        meta, z = load_golden(name)
        scene = golden_scene(meta)
        cams = G.ref_cameras(R, scene)
        depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
        pm = R["PMVO"].PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                            patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
        R["PMVO"].bust_tree = KDTree(data=z["toy_bust"])
        R["PMVO"].scalp_tree = KDTree(data=z["toy_scalp"])
        R["PMVO"].scalp_max = np.max(z["toy_scalp"], axis=0)
        pick = np.arange(0, len(z["points"]), 5)[:24]
        fo, fl, fh, rl = [], [], [], []
        for n in pick:
            p = z["points"][n:n + 1]
            _, so, ml, hc = pm.forward(p)
            fo.append(so.numpy()[0]); fl.append(ml.numpy()[0]); fh.append(hc.numpy()[0])     # noqa: E702
            rl.append(pm.refine(torch.from_numpy(p).type(torch.float), torch.from_numpy(z["refine_ori_in"][n:n + 1])).numpy()[0])
        fpick = np.arange(0, len(z["filter_points_in"]), 37)[:40]
        si, fi, ui = [], [], []
        for n in fpick:
            q = torch.from_numpy(z["filter_points_in"][n:n + 1]).type(torch.float)
            s_, _, f_ = pm.filter_points(q)
            si.append(s_.numpy()[0]); fi.append(f_.numpy()[0]); ui.append(pm.compute_unvisible_points(q).numpy()[0])   # noqa: E702
        out.update({name + "__pick": pick, name + "__fwd_ori": np.array(fo), name + "__fwd_loss": np.array(fl),
                    name + "__fwd_hc": np.array(fh), name + "__refine_loss": np.array(rl), name + "__fpick": fpick,
                    name + "__surface": np.array(si), name + "__filter": np.array(fi), name + "__unvisible": np.array(ui)})
        # how many of these differ from the same points inside their N-point batch (the fixture's own forward)
        same = (np.array(fl) == z["fwd_loss"][pick]) | (np.isnan(fl) & np.isnan(z["fwd_loss"][pick]))
        print("%-12s forward alone != forward in the batch on %d of %d points; refine: %d" % (
            name, int((~same).sum()), len(pick), int((np.array(rl) != z["refine_loss"][pick]).sum())))
    meta = dict(torch=torch.__version__, threads=torch.get_num_threads(), what="each point handed to the reference alone")
    np.savez_compressed(os.path.join(G.OUT, "pmvo_single.npz"), meta=np.array(repr(meta)), **out)
    print("pmvo_single.npz written")


if __name__ == "__main__":
    main()
