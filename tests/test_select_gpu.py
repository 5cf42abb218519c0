"""GPU: the device-side selections between the stages of refine (mh_flag_less, mh_select_rows, mh_segment_heads,
mh_buffers_differ) against the numpy statements they replace (PMVO.py:651-653, 680-693, 705-715) -- exact."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(dev):
    from monohair_amd.pmvo_utils import _ctx_for

    return _ctx_for(dev)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 1023, 1024, 1025, 5000, 70001, 300007])
def test_flag_and_select_rows_equal_numpy(n):
    import torch

    from monohair_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    ctx = _ctx(dev)
    rng = np.random.default_rng(n)
    x = rng.normal(size=n).astype(np.float32)
    if n > 10:
        x[rng.integers(0, n, 5)] = np.nan
    a = rng.normal(size=(n, 3)).astype(np.float32)
    b = rng.normal(size=(n, 3)).astype(np.float32)
    veto = (rng.random(n) < 0.3).astype(np.uint8)
    xd, ad, bd, vd = (torch.from_numpy(t).to(dev) for t in (x, a, b, veto))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    _lib.check(L.mh_flag_less(ctx, _lib.ptr(xd), 0.25, n, _lib.ptr(flags), st))
    want = x < np.float32(0.25)
    assert np.array_equal(flags.cpu().numpy().astype(bool), want)
    scratch = torch.empty(int(L.mh_select_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    cap = 2 * n + 1
    ao = torch.full((cap, 3), -7.0, dtype=torch.float32, device=dev)
    bo = torch.full((cap, 3), -7.0, dtype=torch.float32, device=dev)
    io = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    p = lambda t, o=0: ctypes.c_void_p(t.data_ptr() + o)       # noqa: E731
    _lib.check(L.mh_select_rows(ctx, p(flags), None, 0, n, p(ad), p(bd), p(ao), p(bo), p(io), None, p(cnt), p(scratch),
                                scratch.numel(), st))
    # second selection appended behind the first: flags && !veto, inverted
    _lib.check(L.mh_select_rows(ctx, p(flags), p(vd), 1, n, p(ad), p(bd), p(ao), p(bo), None, p(cnt), p(cnt, 4), p(scratch),
                                scratch.numel(), st))
    c = cnt.cpu().numpy()
    keep2 = ~(want & (veto == 0))
    assert c[0] == want.sum() and c[1] == want.sum() + keep2.sum()
    got_a, got_b, got_i = ao.cpu().numpy(), bo.cpu().numpy(), io.cpu().numpy()
    assert np.array_equal(got_a[:c[1]], np.concatenate([a[want], a[keep2]], 0))
    assert np.array_equal(got_b[:c[1]], np.concatenate([b[want], b[keep2]], 0))
    assert np.array_equal(got_i[:c[0]], np.flatnonzero(want))
    assert (got_a[c[1]:] == -7.0).all()


@pytest.mark.parametrize("n,groups", [(1, 1), (10, 3), (1024, 1024), (4097, 5), (200003, 40000), (50000, 1)])
def test_segment_heads_equal_numpy(n, groups):
    import torch

    from monohair_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    ctx = _ctx(dev)
    rng = np.random.default_rng(n + groups)
    keys = np.sort(rng.integers(0, groups, n)).astype(np.int64) * 7 + 3
    kd = torch.from_numpy(keys).to(dev)
    seg = torch.full((n + 1,), -1, dtype=torch.int32, device=dev)
    heads = torch.empty(n, dtype=torch.int64, device=dev)
    meta = torch.full((2,), -5, dtype=torch.int32, device=dev)
    scratch = torch.empty(int(L.mh_select_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mh_segment_heads(ctx, _lib.ptr(kd), n, _lib.ptr(seg), _lib.ptr(heads), _lib.ptr(meta), _lib.ptr(scratch),
                                  scratch.numel(), st))
    starts = np.flatnonzero(np.concatenate([[True], keys[1:] != keys[:-1]]))
    G, mx = (int(v) for v in meta.cpu().numpy())
    assert G == len(starts) and mx == int(np.diff(np.concatenate([starts, [n]])).max())
    assert np.array_equal(seg.cpu().numpy()[:G + 1], np.concatenate([starts, [n]]))
    assert np.array_equal(heads.cpu().numpy()[:G], keys[starts])


def test_buffers_differ():
    import torch

    from monohair_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    ctx = _ctx(dev)
    a = torch.randn(123457, 3, device=dev)
    a[5, 1] = float("nan")
    b = a.clone()
    flag = torch.full((1,), 9, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mh_buffers_differ(ctx, _lib.ptr(a), _lib.ptr(b), a.numel() * 4, _lib.ptr(flag), st))
    assert int(flag.item()) == 0
    b[123456, 2] += 1.0
    _lib.check(L.mh_buffers_differ(ctx, _lib.ptr(a), _lib.ptr(b), a.numel() * 4, _lib.ptr(flag), st))
    assert int(flag.item()) == 1


def test_voxel_fit_device_equals_voxel_fit():
    import torch

    from monohair_amd import pmvo_utils as U

    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n = 60011
    pts = (rng.normal(size=(n, 3)) * 0.05).astype(np.float32)
    ori = rng.normal(size=(n, 3)).astype(np.float32)
    ref = U.voxel_fit(pts, ori, dev, dense=False)
    vox, vori = U.voxel_fit_device(torch.from_numpy(pts).to(dev), torch.from_numpy(ori).to(dev), n, dev)
    assert np.array_equal(vox, ref["voxels"].cpu().numpy())
    assert np.array_equal(vori, ref["ori"].cpu().numpy())


@pytest.mark.parametrize("n", [1, 63, 4097, 300001])
def test_points_bbox_equals_numpy(n):
    import torch

    from monohair_amd import _lib

    L = _lib.lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(n)
    pts = (rng.normal(size=(n, 3)) * np.array([1.0, 1e-3, 50.0]) + np.array([-3.0, 0.0, 7.0])).astype(np.float32)
    pts[rng.integers(0, n)] = [-0.0, 0.0, -1e-30]
    box = torch.empty(6, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mh_points_bbox(_ctx(dev), _lib.ptr(torch.from_numpy(pts).to(dev)), n, _lib.ptr(box), st))
    got = box.cpu().numpy()
    assert np.array_equal(got[:3], pts.min(0)) and np.array_equal(got[3:], pts.max(0))
