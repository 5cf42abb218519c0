"""Test helper (not a test): run the product's `optimize` / `refine` drivers on the INPUTS stored in a golden file of the
reference's own chunked run (tests/golden/e2e_multichunk.npz) and leave the output files in --out, so that a test can compare
them with the reference's files -- in this process, or as N ranks under torch.distributed.run (gloo ranks sharing the test GPU,
as tests/test_cli_gpu.py does; MH_REFINE_SHARD=1 shards the smoothing loop).

    python tests/golden_drivers.py --out DIR [--what pass,optimize,optimize_exact,refine,refine_exact,refine_headfilter]
"""
import argparse
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def run(out_dir, what=("optimize", "refine", "refine_exact"), device="cuda:0"):
    import torch
    from scipy.spatial import KDTree

    from conftest import GOLDEN, golden_records, golden_scene
    from monohair_amd.pmvo import PMVO, optimize, refine

    z = np.load(os.path.join(GOLDEN, "e2e_multichunk.npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    scene = golden_scene(meta)
    pm = PMVO.from_planes(golden_records(z), scene["depth"].to(device), scene["ori"].to(device), scene["conf"].to(device),
                          scene["mask"].to(device), device=device, patch_size=meta["patch"],
                          visible_threshold=meta["vis_thr"], conf_threshold=meta["thr"])
    scalp = z["toy_scalp"]
    pm.set_head(KDTree(data=z["toy_bust"]), KDTree(data=scalp), np.max(scalp, axis=0))
    fu = z["candidates"][z["filter_index"]]

    def args_for(sub):
        root = os.path.join(out_dir, sub)
        a = types.SimpleNamespace(device=device, output_path=root, save_root=os.path.join(root, "optimize"),
                                  save_path=os.path.join(root, "refine"),
                                  PMVO=types.SimpleNamespace(visible_threshold=meta["vis_thr"]),
                                  data=types.SimpleNamespace(root=root))
        os.makedirs(a.save_path, exist_ok=True)
        return a

    if "pass" in what:
        # the three drivers in sequence on the reference's candidates, arrays handed from one to the next as PMVO.py's caller
        # would: refine() then finds the neighbour table / head votes / .mat pages that optimize() prepared for these points
        import json

        from monohair_amd.pmvo import filter_negative_points

        a = args_for("pass")
        s_idx, s_pts, f_idx = filter_negative_points(z["candidates"].copy(), pm, a)
        sp, so, ml, hc = optimize(s_pts, pm, a)
        np.save(os.path.join(a.save_root, "surface_index.npy"), s_idx)
        np.save(os.path.join(a.save_root, "filter_index.npy"), f_idx)
        refine(sp, so, ml, pm, z["candidates"][:len(f_idx)][f_idx], a, infer_inner=False, threshold=meta["threshold"],
               genrate_ori_only=False, return_dense=False)
        json.dump(pm.last_refine, open(os.path.join(a.output_path, "last_refine.json"), "w"))
    if "optimize" in what:
        # the surface points exactly as the reference's optimize received them (float32 rows of filter_negative_points)
        optimize(z["opt_select_p"].copy(), pm, args_for("run"))
    if "optimize_exact" in what:
        # exactly 10 000 points: `step = N // 5000 + 1` (PMVO.py:566) gives a third chunk of zero points
        optimize(z["opt_select_p"][:meta["exact"]].copy(), pm, args_for("exact"))
    if "refine" in what:
        # from the REFERENCE's optimize outputs, so that both refine stages see identical inputs
        refine(z["opt_select_p"].copy(), z["opt_select_o"].copy(), z["opt_min_loss"].copy(), pm, fu.copy(), args_for("run"),
               infer_inner=False, threshold=meta["threshold"], genrate_ori_only=False, return_dense=False)
    if "refine_exact" in what:
        n = meta["exact"]
        refine(z["opt_select_p"][:n].copy(), z["opt_select_o"][:n].copy(), z["opt_min_loss"][:n].copy(), pm, fu[:3000].copy(),
               args_for("exact"), infer_inner=False, threshold=meta["threshold"], genrate_ori_only=False,
               return_dense=False)
    if "refine_headfilter" in what:
        # tests/golden/e2e_headfilter.npz: a third of the points head-filtered, 40 NaN rows (tools/gen_golden_headfilter.py)
        h = np.load(os.path.join(GOLDEN, "e2e_headfilter.npz"), allow_pickle=False)
        refine(h["in_points"].copy(), h["in_ori"].copy(), h["in_loss"].copy(), pm, h["in_shell"].copy(), args_for("headfilter"),
               infer_inner=False, threshold=meta["threshold"], genrate_ori_only=False, return_dense=False)
    torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--what", default="optimize,refine,refine_exact")
    a = ap.parse_args()
    import torch

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as tdist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("MH_DIST_BACKEND", "nccl") == "nccl":
            tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
        else:
            tdist.init_process_group(backend=os.environ["MH_DIST_BACKEND"])
    local = os.environ.get("MH_DEVICE_OVERRIDE", os.environ.get("LOCAL_RANK", "0"))
    run(a.out, tuple(a.what.split(",")), "cuda:%d" % int(local))


if __name__ == "__main__":
    main()
