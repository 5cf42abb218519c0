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


@pytest.mark.parametrize("rule", [1, 2])
def test_one_point_batches_with_a_forced_product_form_keep_the_inner_sum_order(rule, depth_offsets):
    """Non-default options (round-5 advisor finding): reproject_rule 1 / 2 force one sgemm form for every point, sum_block 32
    keeps ATen's summation rule -- and a [V, 1] sum over the views is an INNER sum whatever the products do.  The kernels
    used to tie the two (the inner-sum order only with reproject_rule 0); the oracle never did.  forward (fused, unfused,
    portable), the method refine and the votes on lone points == oracle with the same options, every value."""
    from test_hip_parity import make_pmvo

    meta, z = load_golden("pmvo_small")
    scene = golden_scene(meta)
    rec = golden_records(z)
    views = scene_views(scene, rec)
    pm = make_pmvo(meta, scene, rec)
    pm.set_option("reproject_rule", rule)
    pm.set_option("sum_block", 32)
    prev = oracle.set_reproject_rule({1: "mid", 2: "chain"}[rule]), oracle.set_sum_block(32)
    try:
        checked = finite = 0
        for n in range(0, len(z["points"]), 3):
            p = z["points"][n:n + 1]
            _, o_o, o_l, o_h = oracle.forward(views, p, meta["patch"], meta["thr"], depth_offsets)
            for variant, fused in ((0, True), (0, False), (1256, True)):
                pm.set_option("search_variant", variant)
                _, o, l, h = pm.forward(p, fused=fused)
                assert np.array_equal(l.cpu().numpy(), o_l, equal_nan=True), (n, variant, fused)
                assert np.array_equal(o.cpu().numpy(), o_o, equal_nan=True) and np.array_equal(h.cpu().numpy(), o_h)
            pm.set_option("search_variant", 0)
            d = z["refine_ori_in"][n:n + 1]
            pm.Compute_Visible_and_Ori(p)
            rl, _ = pm.prj_loss_of(pm._points, torch.from_numpy(d).to(DEV))
            o_rl, _ = oracle.refine_loss(views, p, d, meta["patch"], meta["thr"])
            assert np.array_equal(rl.cpu().numpy(), o_rl, equal_nan=True), n
            q = z["filter_points_in"][n % len(z["filter_points_in"]):][:1]
            got = [t.cpu().numpy() for t in pm._votes(q, (True, True, True, True), meta["vis_thr"])[1]]
            want = oracle.filter_votes(views, q, meta["patch"], meta["thr"], meta["vis_thr"])
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), n
            checked += 1
            finite += int(np.isfinite(o_l).sum())
        assert checked >= 40 and finite >= 10
    finally:
        oracle.set_reproject_rule(*prev[0])
        oracle.set_sum_block(prev[1])
