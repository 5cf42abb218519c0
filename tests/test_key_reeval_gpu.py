"""GPU: a deterministic scene that ENTERS the re-evaluation branch of the search's key body (csrc/pmvo_search.hip: a winner
whose |cos| <= 2^-14 -- or a NaN -- cannot be stated by the integer key; the wave evaluates that view again with the
compare-and-select body, PMVO.py:173-182 either way).  The bench scene takes that branch 0 times in 95 M wave-views, the
adversarial fields of tests/stress_parity.py dedupe to lists the key body never sees: this test builds the case by hand and
asserts, with the library's own counter (mh_debug_key_stats), that the branch ran -- and that every output equals the oracle's.

Construction: for one (point, view, candidate sample) the projected 2D segment direction D is read from the oracle; the 7 x 7
patch of that view around the point's pixel is overwritten with 49 DISTINCT orientations in a fan of +-4.8e-5 rad around the
perpendicular of D (2e-6 rad apart: distinct float32 unit vectors, so no tap is dropped as a duplicate and the list has 49 > 10
taps).  Every tap then has |cos| <= 4.8e-5 < 2^-14 for that candidate: the winner's key is past the valid range."""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def key_stats(pm, reset=True):
    out = (ctypes.c_ulonglong * 4)()
    torch.cuda.synchronize()
    assert pm._L.mh_debug_key_stats(out, 1 if reset else 0) == 0
    return [int(x) for x in out]


def build(V=24, H=240, W=160, patch=7, thr=0.15, seed=5):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list

    scene = synth.make_scene(V, H, W, seed=seed, quantize=False, rings=2)
    cams = cameras_from_list(scene["cams"])
    rec = camera_records(cams)
    pts = synth.candidate_points(res=48, seed=seed, limit=300).astype(np.float32)
    maps = {k: scene[k].cpu().numpy().copy() for k in ("depth", "ori", "conf", "mask")}
    return rec, cams, pts, maps, patch, thr, (V, H, W)


def fan_patch(maps, views_of, rec, pts, patch, thr, offsets, H, W, eps=2e-6, nan_at=None):
    """Overwrite one patch so that one candidate's direction is perpendicular to all of its taps.  Returns (n, v, s)."""
    views = views_of(maps)
    o = oracle.visible_and_ori(views, pts, patch)
    bidx, bval = oracle.topk_views(o["visible"], o["Conf"], 20)
    used = bidx[0:20:2]                                     # base views of the ten ranks forward() evaluates
    samples = oracle.sample_next(views, pts, bidx[0], o["Ori"], offsets)
    D = oracle.reproject_ori(views, pts, samples)            # [V,N,S,2] (row, col)
    hp = patch // 2
    for n in range(len(pts)):
        for v in range(views.V):
            if o["visible"][v, n] == -1.0 or v in used[:, n] or bval[0, n] <= 0:
                continue
            rc, _, oob, _ = oracle.project_points(views.cams[v], pts[n:n + 1], H, W)
            r, c = int(rc[0, 0]), int(rc[0, 1])
            if oob[0] or not (hp <= r < H - hp and hp <= c < W - hp):
                continue
            s = 45
            d = D[v, n, s].astype(np.float64)
            if not np.all(np.isfinite(d)) or np.hypot(*d) < 1e-3:
                continue
            d /= np.hypot(*d)
            perp = np.array([-d[1], d[0]])
            k = 0
            for i in range(-hp, hp + 1):
                for j in range(-hp, hp + 1):
                    a = (k - (patch * patch) // 2) * eps
                    rot = np.array([perp[0] * np.cos(a) - perp[1] * np.sin(a), perp[0] * np.sin(a) + perp[1] * np.cos(a)])
                    maps["ori"][v, r + i, c + j] = rot.astype(np.float32)
                    maps["conf"][v, r + i, c + j] = 0.9                  # every tap eligible (PMVO.py:162,177-182)
                    k += 1
            if nan_at is not None:
                maps["ori"][v, r + nan_at[0], c + nan_at[1]] = np.nan
            return n, v, s
    raise AssertionError("no usable (point, view) pair in this scene")


@pytest.mark.parametrize("nan_at", [None, (1, 2)])
def test_key_body_reevaluation_branch_is_entered_and_exact(nan_at, depth_offsets):
    from monohair_amd.pmvo import PMVO

    rec, cams, pts, maps, patch, thr, (V, H, W) = build()
    views_of = lambda m: oracle.Views(rec, m["depth"], m["ori"], m["conf"], m["mask"])     # noqa: E731
    n, v, s = fan_patch(maps, views_of, rec, pts, patch, thr, depth_offsets, H, W, nan_at=nan_at)
    views = views_of(maps)
    # the construction holds on the final maps: all 49 taps of (v, n) distinct, all within 2^-14 of perpendicular to D[v,n,s]
    o = oracle.visible_and_ori(views, pts, patch)
    bidx, bval = oracle.topk_views(o["visible"], o["Conf"], 20)
    D = oracle.reproject_ori(views, pts, oracle.sample_next(views, pts, bidx[0], o["Ori"], depth_offsets))
    taps = o["Ori_patch"][v, n].astype(np.float64)
    ok = np.all(np.isfinite(taps), axis=1)
    taps[ok] /= np.linalg.norm(taps[ok], axis=1, keepdims=True)
    assert len({t.tobytes() for t in o["Ori_patch"][v, n]}) == patch * patch
    dh = D[v, n, s].astype(np.float64)
    dh /= np.linalg.norm(dh)
    assert np.abs(taps[ok] @ dh).max() < 2.0 ** -14 and o["visible"][v, n] != -1.0
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)       # noqa: E731
    pm = PMVO.from_planes(rec, t(maps["depth"]), t(maps["ori"]), t(maps["conf"]), t(maps["mask"]), device=DEV,
                          patch_size=patch, visible_threshold=1, conf_threshold=thr, camera=cams)
    _, o_ori, o_loss, o_hc, o_ex = oracle.forward(views, pts, patch, thr, depth_offsets, extra=True)
    key_stats(pm)                                                    # reset
    results = {}
    for body, variant, fused in ((1, 0, True), (1, 0, False), (2, 0, True), (0, 1256, True)):
        pm.set_option("search_body", body)
        pm.set_option("search_variant", variant)
        _, ori, loss, hc, ex = pm.forward(pts, extras=True, fused=fused)
        st = key_stats(pm)
        if body == 1:
            assert (pm.search_work(len(pts))[0][v, n].item()) == patch * patch - (0 if nan_at is None else 0)
            assert st[2] > 0, "the key body's re-evaluation branch was not entered"
        else:
            assert st[2] == 0                                        # (the select body and the portable kernel build no keys)
        results[(body, variant, fused)] = tuple(x.cpu().numpy() for x in (ori, loss, hc, ex["best_s"], ex["best_rank"]))
    pm.set_option("search_body", 0)
    pm.set_option("search_variant", 0)
    ref = results[(2, 0, True)]
    for k, got in results.items():
        for a, b in zip(got, ref):
            assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), k     # the bodies agree with each other ...
    for a, b in zip(ref, (o_ori, o_loss, o_hc, o_ex["best_s"], o_ex["best_rank"])):
        assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f"))            # ... and with the oracle, bit for bit
    assert np.isfinite(o_loss[n])
