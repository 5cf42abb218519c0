"""tests/golden/e2e_headfilter.npz -- the branches of refine's smoothing loop that the two end-to-end fixtures do not reach
(run here, from /root/reference, CPU torch):

  * `loss[filter_index] = -1` in PMVO.refine (/root/reference/PMVO.py:91-92) followed by `sub_loss[sub_loss==-1] = 0.5` in
    the loop (:639): no surface point of e2e_small / e2e_multichunk is head-filtered;
  * NaN orientations and losses among the inputs (what `optimize` writes for points whose winning candidate coincides with
    the point: forward goldens have such rows) -- they enter other points' neighbourhoods as NaN cosines.

Inputs: the first 6000 surface points of e2e_multichunk.npz and the reference's own optimize outputs for them, with every
third point pushed outward by 30 % (it then projects outside the hair mask in most views that see it), and 40 rows of NaN
orientation / loss.  Two chunks (5000 + 1000): the second chunk's neighbourhoods reach into the first.

    python tools/gen_golden_headfilter.py
"""
import ast
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


def main():
    from scipy.spatial import KDTree

    torch.manual_seed(0)
    R = import_reference()
    ref = R["PMVO"]
    z = np.load(os.path.join(OUT, "e2e_multichunk.npz"))
    case = ast.literal_eval(str(z["meta"]))
    scene = synth.make_scene(case["V"], case["H"], case["W"], seed=case["seed"], scale=case["scale"],
                             rings=case["rings"], quantize=case["quantize"])
    cams = {}
    for c in scene["cams"]:
        cams[c["file"]] = R["Camera_utils"].Camera(c["ndc_prj"], np.linalg.inv(np.array(c["pose"])), c["file"])
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
    pm = ref.PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                  patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
    assert np.array_equal(np.stack([c.pose.numpy() for c in pm.camera]), z["cam_pose"])      # same camera tensors as the big fixture
    bust, scalp = z["toy_bust"], z["toy_scalp"]
    ref.bust_tree = KDTree(data=bust)
    ref.scalp_tree = KDTree(data=scalp)
    ref.scalp_max = np.max(scalp, axis=0)
    ref.device = "cpu"

    n = 6000
    rng = np.random.default_rng(77)
    pts = z["opt_select_p"][:n].copy()
    ori = z["opt_select_o"][:n].copy()
    loss = z["opt_min_loss"][:n].copy()
    pts[::3] *= np.float32(1.3)
    bad = np.sort(rng.choice(n, 40, replace=False))
    ori[bad] = np.nan
    loss[bad[::2]] = np.nan
    out = dict(in_points=pts.copy(), in_ori=ori.copy(), in_loss=loss.copy(), nan_rows=bad.astype(np.int32))

    tmp = tempfile.mkdtemp(prefix="mh_e2e_hf_")
    args = types.SimpleNamespace(device="cpu", output_path=tmp, save_root=os.path.join(tmp, "optimize"),
                                 save_path=os.path.join(tmp, "refine"),
                                 PMVO=types.SimpleNamespace(visible_threshold=case["vis_thr"]),
                                 data=types.SimpleNamespace(root=tmp))
    os.makedirs(args.save_path, exist_ok=True)
    fu = z["candidates"][z["filter_index"]][:500]
    out["in_shell"] = fu.copy()
    ref.refine(pts.copy(), ori.copy(), loss.copy(), pm, fu.copy(), args, infer_inner=False, threshold=case["threshold"],
               genrate_ori_only=False)
    for k in ("select_p", "select_o", "min_loss", "filter_unvisible", "filter_unvisible_ori"):
        out["ref_" + k] = np.load(os.path.join(tmp, "refine", k + ".npy"))
    shutil.rmtree(tmp)
    l = out["ref_min_loss"]
    print("head-filtered (loss 0.5): chunk 0: %d, chunk 1: %d; NaN losses out: %d; NaN orientation rows out: %d"
          % ((l[:5000] == 0.5).sum(), (l[5000:] == 0.5).sum(), np.isnan(l).sum(), np.isnan(out["ref_select_o"]).any(1).sum()))
    assert (l[:5000] == 0.5).sum() > 50 and (l[5000:] == 0.5).sum() > 10
    np.savez_compressed(os.path.join(OUT, "e2e_headfilter.npz"), meta=np.array(repr(case)), **out)
    print("e2e_headfilter written")


if __name__ == "__main__":
    main()
