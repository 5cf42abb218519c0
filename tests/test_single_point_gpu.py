"""GPU: batches of ONE point (tests/golden/pmvo_single.npz, tools/gen_golden_single.py).  Every projection of a lone point is a
single-column sgemm in the reference, its sums over the views are sums over a contiguous dimension: the last chunk of optimize /
refine when N mod 5000 == 1.  forward, the method refine and the votes against the reference's outputs, every row."""
import os

import numpy as np
import pytest
import torch
from scipy.spatial import KDTree

import oracle
from conftest import GOLDEN, golden_records, golden_scene, load_golden, scene_views

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("name", ["pmvo_small", "pmvo_quant"])
def test_batches_of_one_point_equal_the_reference(name, depth_offsets):
    from test_hip_parity import make_pmvo

    meta, z = load_golden(name)
    scene = golden_scene(meta)
    rec = golden_records(z)
    views = scene_views(scene, rec)
    pm = make_pmvo(meta, scene, rec)
    scalp = z["toy_scalp"]
    pm.set_head(KDTree(data=z["toy_bust"]), KDTree(data=scalp), np.max(scalp, axis=0))
    s = np.load(os.path.join(GOLDEN, "pmvo_single.npz"))
    g = lambda k: s[name + "__" + k]                                                    # noqa: E731
    differs = 0
    for i, n in enumerate(g("pick")):
        p = z["points"][n:n + 1]
        for variant, fused in ((0, True), (0, False), (1256, True)):
            pm.set_option("search_variant", variant)
            _, o, l, h = pm.forward(p, fused=fused)
            o, l, h = o.cpu().numpy(), l.cpu().numpy(), h.cpu().numpy()
            assert np.array_equal(o[0], g("fwd_ori")[i], equal_nan=True), (i, variant, fused)
            assert np.array_equal(l[0], g("fwd_loss")[i], equal_nan=True) and h[0] == g("fwd_hc")[i], (i, variant, fused)
        pm.set_option("search_variant", 0)
        differs += not np.array_equal(l[0], z["fwd_loss"][n], equal_nan=True)
        rl = pm.refine(torch.from_numpy(p).to(DEV).float(), torch.from_numpy(z["refine_ori_in"][n:n + 1]).to(DEV)).cpu().numpy()
        assert np.array_equal(rl[0], g("refine_loss")[i], equal_nan=True), (i, rl[0], g("refine_loss")[i])
        o_rl, _ = oracle.refine_loss(views, p, z["refine_ori_in"][n:n + 1], meta["patch"], meta["thr"])
        assert rl[0] == -1 or np.array_equal(rl[0], o_rl[0], equal_nan=True)
    assert differs > 5
    for i, n in enumerate(g("fpick")):
        q = torch.from_numpy(z["filter_points_in"][n:n + 1]).to(DEV).float()
        surf, _, filt = pm.filter_points(q)
        unv = pm.compute_unvisible_points(q)
        assert bool(surf[0]) == g("surface")[i] and bool(filt[0]) == g("filter")[i] and bool(unv[0]) == g("unvisible")[i]
    # the batch-independent option keeps one form: a lone point then gives its answer inside any larger batch of rounds 1-4
    pm.set_option("reproject_rule", 1)
    pm.set_option("sum_block", 0)
    n = int(g("pick")[0])
    _, o1, l1, _ = pm.forward(z["points"][n:n + 1])
    _, oN, lN, _ = pm.forward(z["points"])
    pm.set_option("reproject_rule", 0)
    pm.set_option("sum_block", 32)
    assert torch.equal(torch.nan_to_num(l1, nan=-7.0)[0], torch.nan_to_num(lN, nan=-7.0)[n])
