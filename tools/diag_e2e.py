#!/usr/bin/env python
"""Why do some points of optimize() differ from the reference's e2e run?  Prints the match rate and, for the mismatches,
whether the point's base-view ranking has ties at a used rank and how many points share its rank-0 base view."""
import ast, os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, golden_records, golden_scene
from monohair_amd.pmvo import PMVO
DEV = "cuda:0"
z = np.load(os.path.join(GOLDEN, "e2e_small.npz"), allow_pickle=False)
meta = ast.literal_eval(str(z["meta"]))
scene = golden_scene(meta)
pm = PMVO.from_planes(golden_records(z), scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                      scene["mask"].to(DEV), device=DEV, patch_size=meta["patch"], visible_threshold=meta["vis_thr"],
                      conf_threshold=meta["thr"])
pts = z["opt_select_p"]
p, o, l, hc, ex = pm.forward(pts, extras=True)
o, l = o.cpu().numpy(), l.cpu().numpy()
same = ((l == z["opt_min_loss"]) | (np.isnan(l) & np.isnan(z["opt_min_loss"]))) & np.all((o == z["opt_select_o"]) | np.isnan(z["opt_select_o"]), 1)
print("points", len(pts), "identical", same.mean())
bv = ex["base_val"].cpu().numpy(); bi = ex["base_idx"].cpu().numpy()
used = bv[0:20:2]
ties = np.zeros(len(pts), bool)
for r in range(0, 20):
    if r + 1 < 20:
        ties |= (bv[r] == bv[r + 1]) & (bv[r] > 0)
cnt0 = np.bincount(bi[0], minlength=pm.num_view)
single = np.zeros(len(pts), bool)
for r in range(0, 20, 2):
    c = np.bincount(bi[r], minlength=pm.num_view)
    single |= (c[bi[r]] == 1) & (bv[r] > 0)
bad = ~same
print("mismatches", bad.sum(), "with positive-value ties in top-20", (bad & ties).sum(), "with a base view owning one point at some rank", (bad & single).sum())
print("all points: ties", ties.mean(), "single", single.mean())
d = np.abs(l - z["opt_min_loss"])
print("loss diff on mismatches: max %.3g median %.3g" % (np.nanmax(d[bad]) if bad.any() else 0, np.nanmedian(d[bad]) if bad.any() else 0))
print("best_rank hist of mismatches", np.bincount(ex["best_rank"].cpu().numpy()[bad], minlength=10))
nvis = (pm.visible.cpu().numpy() != -1).sum(0)
print("visible views: mismatches mean %.1f, all mean %.1f" % (nvis[bad].mean() if bad.any() else 0, nvis.mean()))
