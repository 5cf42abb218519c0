"""GPU: shapes off the beaten path -- large patches (more taps than lanes), many views (several cascade blocks),
views < 20 (the reference raises), points outside every frustum, zero-confidence maps (NaN losses)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(V, H, W, patch, thr=0.15, quantize=False, seed=2, rings=1):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    scene = synth.make_scene(V, H, W, seed=seed, quantize=quantize, rings=rings)
    rec = camera_records(cameras_from_list(scene["cams"]))
    pm = PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                          scene["mask"].to(DEV), device=DEV, patch_size=patch, visible_threshold=1, conf_threshold=thr)
    views = oracle.Views(rec, scene["depth"].numpy(), scene["ori"].numpy(), scene["conf"].numpy(), scene["mask"].numpy())
    return scene, pm, views


def check_forward(pm, views, pts, patch, thr):
    from monohair_amd.pmvo import depth_offsets

    for fused in (True, False):
        _, ori, loss, hc = pm.forward(pts, fused=fused)
        _, o_ori, o_loss, o_hc = oracle.forward(views, pts, patch, thr, depth_offsets(90))
        assert np.array_equal(loss.cpu().numpy(), o_loss, equal_nan=True)
        assert np.array_equal(ori.cpu().numpy(), o_ori, equal_nan=True)
        assert np.array_equal(hc.cpu().numpy(), o_hc)
    return o_loss


@pytest.mark.parametrize("patch", [1, 9, 11])
def test_patch_sizes_beyond_one_wave(patch):
    """patch 9 / 11 have 81 / 121 taps: more than the 64 lanes of the wave that prepares a tap list"""
    from monohair_amd import synth

    scene, pm, views = build(24, 200, 120, patch)
    pts = synth.candidate_points(res=32, seed=3, limit=150)
    loss = check_forward(pm, views, pts, patch, 0.15)
    assert np.isfinite(loss).sum() > 30
    o = oracle.visible_and_ori(views, pts, patch)
    pm.Compute_Visible_and_Ori(pts)
    assert np.array_equal(pm.Conf_patch.cpu().numpy(), o["Conf_patch"])
    surf, _, filt = pm.filter_points(pts)
    o_s, o_f, _, _ = oracle.filter_votes(views, pts, patch, 0.15, 1.0)
    assert np.array_equal(surf.cpu().numpy(), o_s) and np.array_equal(filt.cpu().numpy(), o_f)


@pytest.mark.parametrize("V", [120, 300])
def test_many_views_cascade_blocks(V):
    """V = 120 (BASELINE config 5): eight 16-view blocks in the cascade sum, two rounds of lanes-over-views;
    V = 300 (an unstrided ~300-frame capture): the cascade's third level takes over at 256 rows"""
    from monohair_amd import synth

    scene, pm, views = build(V, 96, 64, 3, rings=3)
    pts = synth.candidate_points(res=32, seed=4, limit=120)
    check_forward(pm, views, pts, 3, 0.15)
    pm.Compute_Visible_and_Ori(pts)
    dirs = torch.tensor([[0.0, -1.0, 0.0]], device=DEV).repeat(len(pts), 1)
    loss, _ = pm.prj_loss_of(pm._points, dirs)
    o_loss, _ = oracle.refine_loss(views, pts, np.tile([[0.0, -1.0, 0.0]], (len(pts), 1)), 3, 0.15)
    assert np.array_equal(loss.cpu().numpy(), o_loss, equal_nan=True)
    surf, _, filt = pm.filter_points(pts)
    unv = pm.compute_unvisible_points(pts)
    o_s, o_f, o_u, _ = oracle.filter_votes(views, pts, 3, 0.15, 1.0)
    assert np.array_equal(surf.cpu().numpy(), o_s) and np.array_equal(filt.cpu().numpy(), o_f)
    assert np.array_equal(unv.cpu().numpy(), o_u)


def test_fewer_than_twenty_views_fails_like_the_reference():
    """torch.topk(Conf, 20, dim=0) raises for V < 20 (PMVO.py:341); so does mh_topk_views"""
    from monohair_amd import _lib, synth

    scene, pm, views = build(8, 64, 48, 3)
    pts = synth.candidate_points(res=32, seed=5, limit=20)
    with pytest.raises(_lib.MhError, match="views < 20"):
        pm.forward(pts)
    pm.Compute_Visible_and_Ori(pts)                      # everything that does not rank base views still works
    assert pm.visible.shape == (8, 20)


def test_points_outside_and_dead_maps():
    from monohair_amd import synth

    scene, pm, views = build(24, 96, 64, 5)
    far = np.array([[9.0, 9.0, 9.0], [0.0, 0.0, 5.0], [-3.0, 0.1, 0.0]])
    pts = np.concatenate([synth.candidate_points(res=32, seed=6, limit=40), far])
    loss = check_forward(pm, views, pts, 5, 0.15)
    assert np.isnan(loss[-3])                            # outside every frustum: no visible view, 0/0 (PMVO.py:201)
    # all-zero confidence: every tap clamps to 1e-6 (PMVO.py:372,376), 'positive' never fires
    scene["conf"].zero_()
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    rec = camera_records(cameras_from_list(scene["cams"]))
    pm0 = PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                           scene["mask"].to(DEV), device=DEV, patch_size=5, visible_threshold=1, conf_threshold=0.15)
    views0 = oracle.Views(rec, scene["depth"].numpy(), scene["ori"].numpy(), scene["conf"].numpy(),
                          scene["mask"].numpy())
    check_forward(pm0, views0, pts[:40], 5, 0.15)


def test_bad_arguments_are_reported_not_crashed():
    from monohair_amd import _lib, synth

    scene, pm, views = build(24, 64, 48, 3)
    with pytest.raises(_lib.MhError):
        pm._side = 4                                      # the C ABI takes odd window sides only (the class maps even
        pm.forward(synth.candidate_points(res=32, seed=6, limit=10))    # patch sizes to the reference's odd window)
    pm._side = 13                                         # larger than the instantiated kernels (documented limit: 11)
    with pytest.raises(_lib.MhError):
        pm.Compute_Visible_and_Ori(synth.candidate_points(res=32, seed=6, limit=10))


def test_medoid_groups_larger_than_lds():
    """a voxel / neighbourhood with more members than the 4096 unit vectors the kernel keeps in LDS (never the case
    for a real capture, but it must not fail): the staged path gives the oracle's medoid too"""
    import oracle
    from monohair_amd.pmvo_utils import compute_points_similarity, voxel_fit

    rng = np.random.default_rng(0)
    ori = rng.normal(size=(2, 4500, 3)).astype(np.float32)
    got = compute_points_similarity(torch.from_numpy(ori).to(DEV)).cpu().numpy()
    want, _ = oracle.medoid_dense(ori)
    assert np.array_equal(got, want)
    p = np.concatenate([rng.normal(0, 0.0002, (6000, 3)), rng.uniform(-0.1, 0.1, (500, 3))])   # 6000 points in one voxel
    o = rng.normal(size=(6500, 3)).astype(np.float32)
    res = voxel_fit(p.copy(), o.copy(), DEV)
    occ, ori_d = oracle.voxel_fit(p.copy(), o.copy(), [-0.32, -0.32, -0.24], 0.005 / 2, [256, 256, 192])
    assert np.array_equal(res["occ"], occ)
    assert np.array_equal(res["ori_dense"].astype(np.float32), ori_d.astype(np.float32))


def test_voxel_group_is_numpys_p2v_on_awkward_points():
    """mh_voxel_group == the reference's grouping (p2v in float64 + dict of lists in point order, PMVO_utils.py:386-404,
    PMVO.py:697-715) on points exactly between voxels (round half to even), outside the grid, non-finite, float64 input."""
    import ctypes

    from monohair_amd import _lib
    from monohair_amd.pmvo_utils import GRID_RESOLUTION, VOXEL_MIN, VOXEL_SIZE, _ctx_for, p2v

    rng = np.random.default_rng(5)
    g = np.asarray(GRID_RESOLUTION).astype(np.int64)
    vmin = np.asarray(VOXEL_MIN, np.float64)
    for dtype in (np.float32, np.float64):
        n = 20000
        pts = rng.uniform(-0.4, 0.4, (n, 3))
        k = rng.integers(0, 255, (4000, 3))
        half = (vmin + (k + 0.5) * VOXEL_SIZE) * np.array([1.0, -1.0, -1.0])       # ties of the rounding (before the y,z flip)
        pts[:4000] = half
        pts[4000:4010] = [[np.nan, 0, 0], [0, np.inf, 0], [0, 0, -np.inf], [1e30, 0, 0], [-1e30, 0, 0], [0, 3e9, 0],
                          [0, -3e9, 0], [0, 0, 5.0], [0, 0, -5.0], [np.nan, np.nan, np.nan]]
        pts = pts.astype(dtype)
        ori = rng.normal(size=(n, 3)).astype(np.float32)
        ori[::7, 1] = 0.0
        ori[3::11, 1] = -0.0
        with np.errstate(all="ignore"):
            x, y, z = p2v(pts.copy(), vmin, VOXEL_SIZE, g)
        key = (x.astype(np.int64) * int(g[1]) + y.astype(np.int64)) * int(g[2]) + z.astype(np.int64)
        want_order = np.argsort(key, kind="stable")
        want_ori = ori.copy()
        want_ori[want_ori[:, 1] > 0] *= -1
        L = _lib.lib()
        pd = torch.from_numpy(pts).to(DEV)
        od = torch.from_numpy(ori).to(DEV)
        ks = torch.empty(n, dtype=torch.int64, device=DEV)
        order = torch.empty(n, dtype=torch.int32, device=DEV)
        osort = torch.empty((n, 3), dtype=torch.float32, device=DEV)
        scratch = torch.empty(int(L.mh_voxel_group_scratch_bytes(n)), dtype=torch.uint8, device=DEV)
        dims = np.ascontiguousarray(g, dtype=np.int32)
        _lib.check(L.mh_voxel_group(_ctx_for(DEV), _lib.ptr(pd), 1 if dtype == np.float64 else 0, _lib.ptr(od), n,
                                    vmin.ctypes.data_as(ctypes.c_void_p), float(VOXEL_SIZE),
                                    dims.ctypes.data_as(ctypes.c_void_p), _lib.ptr(scratch), scratch.numel(), _lib.ptr(ks),
                                    _lib.ptr(order), _lib.ptr(osort), _lib.stream_ptr()))
        torch.cuda.synchronize()
        assert np.array_equal(ks.cpu().numpy(), key[want_order]), dtype
        assert np.array_equal(order.cpu().numpy(), want_order), dtype
        assert np.array_equal(osort.cpu().numpy(), want_ori[want_order]), dtype


def test_even_patch_size_uses_the_reference_tap_window():
    """range(-(size//2), size//2+1) (PMVO.py:494-495): patch_size 4 samples the same 5 x 5 taps as patch_size 5"""
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    V, H, W = 24, 120, 90
    scene = synth.make_scene(V, H, W, seed=2)
    recs = camera_records(cameras_from_list(scene["cams"]))
    pts = synth.candidate_points(res=32, seed=1)[:400]
    out = []
    for ps in (4, 5):
        pm = PMVO.from_planes(recs, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                              scene["mask"].to(DEV), device=DEV, patch_size=ps, conf_threshold=0.15)
        assert pm.patch_size == ps
        pm.Compute_Visible_and_Ori(pts)
        out.append((pm.Ori_patch.clone(), pm.Conf_patch.clone(), [t.clone() for t in pm.forward(pts)[1:]],
                    [t.clone() for t in pm.filter_points(pts)]))
    assert out[0][0].shape[2] == 25 and torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2] + out[0][3], out[1][2] + out[1][3]):
        assert torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b.float(), nan=-7.0))


def test_topk_tie_order_is_torchs_on_adversarial_columns():
    """mh_topk_views (default order) == torch.topk on the CPU, index for index: heavy ties, NaNs, sorted / organ-pipe /
    median-of-three-killer columns that push the library's selection into its heap fall-back, 20..1024 views."""
    import ctypes

    from monohair_amd import _lib
    from monohair_amd.pmvo import PMVO

    rng = np.random.default_rng(8)

    def killer(n):
        v, k = np.zeros(n, np.float32), n // 2
        for i in range(1, k + 1):
            if i % 2 == 1:
                v[i - 1], v[i] = i, k + i
            v[k + i - 1] = 2 * i
        return v

    for V in (20, 21, 37, 60, 63, 64, 65, 127, 128, 129, 300, 512, 1000, 1024):
        N = 96
        cols = []
        for n in range(N):
            m = n % 6
            if m == 0:
                c = rng.choice(np.round(rng.random(int(rng.integers(1, 6))), 2), size=V)
            elif m == 1:
                c = np.arange(V) / V
            elif m == 2:
                c = np.arange(V)[::-1] / V
            elif m == 3:
                c = killer(V) / (2 * V)
            elif m == 4:
                c = np.where(np.arange(V) % 2, np.arange(V), V - np.arange(V)) / V
            else:
                c = np.where(rng.random(V) < 0.5, 1.0, rng.random(V))
            c = np.clip(np.asarray(c, np.float32), 0.0, 1.0)
            if n % 17 == 0:
                c[rng.integers(0, V, 2)] = np.nan
            cols.append(c)
        conf = np.stack(cols, 1).astype(np.float32)                     # [V,N]
        vis = np.ones((V, N), np.float32)
        vis[rng.random((V, N)) < 0.2] = -1.0                             # invisible views: value 0 (more ties)
        want_i, want_v = oracle.topk_views(vis, conf, 20)
        t = torch.where(torch.from_numpy(vis) < 1, torch.from_numpy(conf) * torch.clamp(torch.from_numpy(vis), min=0),
                        torch.from_numpy(conf))
        assert np.array_equal(torch.topk(t, 20, dim=0).indices.numpy(), want_i)       # the oracle really is torch's order
        ctx = ctypes.c_void_p()
        L = _lib.lib()
        _lib.check(L.mh_ctx_create(0, ctypes.byref(ctx)))
        _lib.check(L.mh_ctx_alloc_views(ctx, V, 4, 4))
        vd, cd = torch.from_numpy(vis).to(DEV), torch.from_numpy(conf).to(DEV)
        oi = torch.empty((20, N), dtype=torch.int32, device=DEV)
        ov = torch.empty((20, N), dtype=torch.float32, device=DEV)
        for order in (0, 0 | (8 << 8), 0 | (16 << 8)):       # the wave form: 4 (default), 8 and 16 points per workgroup
            _lib.check(L.mh_ctx_set_option(ctx, b"topk_order", order))
            oi.fill_(-1)
            _lib.check(L.mh_topk_views(ctx, _lib.ptr(vd), _lib.ptr(cd), N, _lib.ptr(oi), _lib.ptr(ov), _lib.stream_ptr()))
            torch.cuda.synchronize()
            assert np.array_equal(oi.cpu().numpy(), want_i), (V, order)
            assert np.array_equal(ov.cpu().numpy(), want_v, equal_nan=True), (V, order)
        L.mh_ctx_destroy(ctx)


def test_upload_ring_wraps_and_changes_size_with_many_calls_in_flight():
    """forward() uploads its host chunk through a ring of 32 pinned slots per launch stream and mh_upload_pinned: more calls
    than slots without a synchronisation in between (the ring wraps while copies are in flight), chunk sizes that change from
    call to call and one that exceeds the slab (re-allocation) must all read the right points: every result is compared with
    the same call made on a device tensor."""
    from monohair_amd import _lib, synth

    scene, pm, views = build(24, 96, 80, 3)
    cand = synth.candidate_points(res=32, seed=1)
    rng = np.random.default_rng(0)
    sizes = [int(rng.integers(1, 300)) for _ in range(80)] + [9000] + [17, 230]
    sizes[40] = 0
    chunks = [np.ascontiguousarray(cand[rng.choice(len(cand), n, replace=n > len(cand))]) for n in sizes]   # float64 in
    st = pm.side_streams(2)
    outs = []
    for i, c in enumerate(chunks):                              # no synchronisation: 83 calls over 2 x 32 slots
        with torch.cuda.stream(st[i % 2]):
            outs.append(pm.forward(c)[1:])
    torch.cuda.synchronize()
    for c, (o, l, h) in zip(chunks, outs):
        d = torch.from_numpy(c.astype(np.float32)).to(DEV)
        _, o2, l2, h2 = pm.forward(d)
        assert torch.equal(o, o2) or (torch.isnan(o) == torch.isnan(o2)).all() and torch.equal(torch.nan_to_num(o), torch.nan_to_num(o2))
        assert np.array_equal(l.cpu().numpy(), l2.cpu().numpy(), equal_nan=True) and torch.equal(h, h2)
    L = _lib.lib()
    assert L.mh_upload_async(pm._ctx, None, None, 16, None) != 0 and b"mh_upload_async" in L.mh_last_error()
    assert L.mh_upload_async(pm._ctx, None, None, 0, None) == 0
    assert L.mh_upload_pinned(pm._ctx, None, None, 16, None) != 0 and b"mh_upload_pinned" in L.mh_last_error()
    assert L.mh_upload_pinned(pm._ctx, None, None, 0, None) == 0


def test_upload_pinned_kernel_and_copy_engine_forms_move_the_same_bytes():
    """mh_upload_pinned: up to 1 MiB (4-byte multiples) as a kernel that reads the page-locked buffer, anything else as
    hipMemcpyAsync -- sizes on both sides of the switch, odd sizes, unaligned offsets; mh_buffers_differ agrees with torch."""
    import ctypes

    from monohair_amd import _lib

    scene, pm, views = build(20, 48, 40, 3)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    host = torch.empty(3 << 20, dtype=torch.uint8, pin_memory=True)
    host.numpy()[:] = np.random.default_rng(1).integers(0, 256, host.numel(), dtype=np.uint8)
    for nbytes, off in ((4, 0), (60000, 0), (1 << 20, 0), ((1 << 20) + 4, 0), (3 << 20, 0), (1023, 0), (4096, 1), (4096, 8)):
        nbytes = min(nbytes, host.numel() - off)
        dev = torch.zeros(nbytes + 8, dtype=torch.uint8, device=DEV)
        _lib.check(L.mh_upload_pinned(pm._ctx, host.data_ptr() + off, dev.data_ptr(), nbytes, st), "mh_upload_pinned")
        torch.cuda.synchronize()
        got = dev.cpu().numpy()
        assert np.array_equal(got[:nbytes], host.numpy()[off:off + nbytes]) and not got[nbytes:].any(), (nbytes, off)
    a = torch.randn(70001, 3, device=DEV)
    flag = torch.empty(1, dtype=torch.int32, device=DEV)
    for change in (False, True):
        b = a.clone()
        if change:
            b[70000, 2] = b[70000, 2] + 1
        _lib.check(L.mh_buffers_differ(pm._ctx, _lib.ptr(a), _lib.ptr(b), a.numel() * 4, _lib.ptr(flag), st))
        assert bool(flag.item()) == change


def test_tap_plane_respects_its_budget(monkeypatch):
    """The plane of ready-made taps doubles the resident map memory: contexts only keep it below tap_plane_max_mb (environment
    MH_TAP_PLANE_MAX_MB at creation); with or without it forward() returns the same bits."""
    from monohair_amd import synth

    cand = synth.candidate_points(res=32, seed=1)[:257]
    scene, pm, views = build(24, 96, 80, 5)
    ref = [t.cpu().numpy() for t in pm.forward(cand)[1:]]
    monkeypatch.setenv("MH_TAP_PLANE_MAX_MB", "0")
    scene2, pm2, _ = build(24, 96, 80, 5)
    got = [t.cpu().numpy() for t in pm2.forward(cand)[1:]]
    assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got))
    with pytest.raises(_lib_error()):
        pm2.set_option("tap_plane_max_mb", -1)
    from monohair_amd import _lib as _l

    with pytest.raises(_lib_error()):         # a lab key through the supported entry point is refused, and says where it lives
        _l.check(pm2._L.mh_ctx_set_option(pm2._ctx, b"search_variant", 7), "mh_ctx_set_option")
    assert b"mh_pmvo_lab.h" in pm2._L.mh_last_error()
    pm2.set_option("search_variant", 7)           # ... but PMVO.set_option routes lab keys to mh_ctx_set_lab_option
    pm2.set_option("search_variant", 0)


def _lib_error():
    from monohair_amd._lib import MhError

    return MhError


@pytest.mark.parametrize("N", [0, 29, 59, 3000, 3001])
def test_filter_negative_points_chunking_quirks(N, tmp_path):
    """filter_negative_points (PMVO.py:535-557) walks `step` = 30 or 31 pieces of N // 30 points: with N < 30
    every piece is empty (nothing is covered), with N = 59 only the first 31 points are looked at, with N a multiple of 30
    there are exactly 30 pieces.  The masks returned cover what the reference covers, and equal the oracle's votes there."""
    import types

    from monohair_amd import synth
    from monohair_amd.pmvo import filter_negative_points

    scene, pm, views = build(24, 120, 90, 3)
    cand = synth.candidate_points(res=32, seed=3)
    scale = np.ones(len(cand))
    scale[::3], scale[1::3] = 1.05, 0.93                       # surface, outside and inside points
    pts = (cand * scale[:, None])[:N]
    args = types.SimpleNamespace(device=DEV)
    surf, spts, filt = filter_negative_points(pts, pm, args)
    step = 30 if N % 30 == 0 else 31
    covered = min(N, step * (N // 30))
    assert len(surf) == covered and len(filt) == covered and surf.dtype == np.bool_ and filt.dtype == np.bool_
    assert spts.dtype == np.float32 and spts.shape == (int(surf.sum()), 3)
    if covered:
        o_s, o_f, _, _ = oracle.filter_votes(views, pts[:covered], 3, 0.15, 1.0)
        assert np.array_equal(surf, o_s) and np.array_equal(filt, o_f)
        assert np.array_equal(spts, pts[:covered][o_s].astype(np.float32))
        assert 0 < o_s.sum() < covered or covered < 40


@pytest.mark.parametrize("patch,quantize", [(5, False), (3, True), (9, False)])
def test_votes_lane_per_point_kernel_equals_wave_per_point_and_oracle(patch, quantize):
    """Round 6: mh_filter_points launches of >= 4096 points vote with lane = point (mh_filter_rows_kernel); the trailing
    (len mod 32) rows of every batch and one-point batches stay with the wave-per-point kernel.  Same bits as the wave-per-point
    kernel for every row (lab switch filter_rows 0) in one batch, in batches that do not divide the launch, in a slice of a
    longer run, with a one-point last batch -- and equal to the oracle batch by batch."""
    import ctypes

    from monohair_amd import _lib, synth

    scene, pm, views = build(24, 96, 80, patch, quantize=quantize)
    cand = synth.candidate_points(res=48, seed=3)
    rng = np.random.default_rng(patch)
    pts = (cand[rng.choice(len(cand), 12001, replace=len(cand) < 12001)] * rng.uniform(0.95, 1.05, (12001, 1))).astype(np.float32)
    d = torch.from_numpy(pts).to(DEV)
    L = _lib.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def votes(lo, n, batch, row0, total, rows):
        pm.set_option("filter_rows", rows)
        outs = [torch.full((n,), 7, dtype=torch.uint8, device=DEV) for _ in range(4)]
        _lib.check(L.mh_filter_points(pm._ctx, ctypes.c_void_p(d.data_ptr() + lo * 12), n, pm._side, 0.15, 1.0,
                                      *[_lib.ptr(o) for o in outs], batch, row0, total, st), "mh_filter_points")
        return [o.cpu().numpy() for o in outs]

    cases = [(0, 12001, 0, 0, 0),            # one batch of 12001 = 375 * 32 + 1
             (0, 12001, 1777, 0, 12001),     # batches that do not divide the launch, last one of 1339 rows
             (3000, 6000, 1000, 3000, 12001),  # a slice of a longer run (one-point LAST batch outside the slice)
             (5000, 7001, 1000, 5000, 12001),  # ... and the slice that holds that one-point batch
             (0, 4096, 4096, 0, 4096)]        # tails of length 0: only the lane-per-point kernel runs
    for lo, n, batch, row0, total in cases:
        a, b = votes(lo, n, batch, row0, total, 1), votes(lo, n, batch, row0, total, 0)
        for k in range(4):
            assert np.array_equal(a[k], b[k]) and a[k].max() <= 1, (lo, n, batch, k, int((a[k] != b[k]).sum()))
        # the oracle, batch by batch
        bsz = batch if batch else n
        tot = total if batch else n
        want = [np.zeros(n, bool) for _ in range(4)]
        for s0 in range(row0 // bsz * bsz, row0 + n, bsz):
            e0 = min(s0 + bsz, tot)
            got = oracle.filter_votes(views, pts[lo + (s0 - row0):lo + (e0 - row0)] if s0 >= row0 else
                                      pts[lo - (row0 - s0):lo + (e0 - row0)], patch, 0.15, 1.0)
            a0, a1 = max(s0, row0) - row0, min(e0, row0 + n) - row0
            off = max(s0, row0) - s0
            for k in range(4):
                want[k][a0:a1] = got[k][off:off + (a1 - a0)]
        for k in range(4):
            assert np.array_equal(a[k].astype(bool), want[k]), (lo, n, batch, k)
    pm.set_option("filter_rows", 1)
    # rows taken in another order (mh_filter_points_ordered: a random permutation, and the cell order the drivers use): the same
    # votes row for row
    from monohair_amd.pmvo_utils import spatial_order

    lo, n, batch, row0, total = cases[1]
    ref = votes(lo, n, batch, row0, total, 1)
    for order in (torch.from_numpy(rng.permutation(n).astype(np.int32)).to(DEV), spatial_order(d[lo:lo + n].contiguous())):
        assert sorted(order.cpu().numpy().tolist()) == list(range(n))
        outs = [torch.full((n,), 7, dtype=torch.uint8, device=DEV) for _ in range(4)]
        _lib.check(L.mh_filter_points_ordered(pm._ctx, ctypes.c_void_p(d.data_ptr() + lo * 12), n, pm._side, 0.15, 1.0,
                                              *[_lib.ptr(o) for o in outs], batch, row0, total, _lib.ptr(order), st))
        assert all(np.array_equal(o.cpu().numpy(), r) for o, r in zip(outs, ref))
