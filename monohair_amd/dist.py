"""Multi-GPU plumbing of the PMVO path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by POINTS, not by views: a point's result needs every view but no other point, every GPU
holds all packed views (2.5 GB at 60 x 1080p, nothing against 288 GB), so an iteration has no collective at
all and results are bit-identical to the single-GPU run.  Exactly two kinds of exchange exist:
  * map_chunks: independent chunks dealt round-robin, one all_gather of the per-chunk results at the end;
  * voxel_fit_reduced: every rank fits a disjoint x-slab of voxels into a zero SLAB-sized buffer; ONE exchange over
    xGMI assembles the shared 3D orientation/occupancy volume on rank 0 (the only rank that holds it dense).  Because
    ownership is disjoint, every peer sends its slab straight to the root (RCCL send/recv), 1/N of the bytes of a dense
    reduce per link.  Which binding issues those sends is MH_VOLUME_EXCHANGE:
      "torch" (default)  torch.distributed.batch_isend_irecv -- ncclSend/ncclRecv through torch's own RCCL binding;
      "capi"             mh_volume_gather of the C ABI (librccl bound by hand in csrc/capi.cpp) -- opt-in until it has
                         run across devices; tested with several ranks sharing one GPU through tests/fake_rccl.cpp;
      "dense"            the dense ncclReduce(sum) of a full volume per rank (x + 0 is exact: the same volume).
    All three give the single-GPU volume bit for bit.
"""
import numpy as np
import torch


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def rank():
    d = _dist()
    return d.get_rank() if d else 0


def world():
    d = _dist()
    return d.get_world_size() if d else 1


def barrier():
    d = _dist()
    if d:
        d.barrier()


def _comm_device(device):
    """collectives run on the GPU with RCCL; the gloo backend (CPU tests, single-GPU multi-rank tests) stages on the host"""
    d = _dist()
    return device if (d and d.get_backend() == "nccl") else "cpu"


def owner(i, n_ranks=None):
    """rank that processes chunk i"""
    return i % (world() if n_ranks is None else n_ranks)


def map_chunks(chunks, fn, device, empty, after=None, with_index=False):
    """Apply fn to every chunk (each rank takes chunks i with i % world == rank) and return the list of all
    results in chunk order on every rank.  fn returns a 2-D tensor on `device`; `empty()` gives a [0,C] one.
    after(): called once all chunks have been issued and before any result is read (e.g. to join side streams).
    with_index: fn(chunk, i) instead of fn(chunk)."""
    d = _dist()
    w, r = world(), rank()
    call = (lambda c, i: fn(c, i)) if with_index else (lambda c, i: fn(c))
    mine = {i: call(c, i) if len(c) else empty() for i, c in enumerate(chunks) if owner(i, w) == r}
    if after is not None:
        after()
    if not d:
        return [mine[i] for i in range(len(chunks))]
    # one all_gather of padded per-rank buffers (sizes are known on every rank: they are the chunk lengths)
    lens = [len(c) for c in chunks]
    cols = empty().shape[1]
    dt = empty().dtype
    per_rank = [sum(lens[i] for i in range(len(chunks)) if owner(i, w) == k) for k in range(w)]
    cap = max(per_rank) if per_rank else 0
    cdev = _comm_device(device)
    buf = torch.zeros((max(cap, 1), cols), dtype=dt, device=cdev)
    if mine:
        cat = torch.cat([mine[i] for i in sorted(mine)], 0)
        buf[:cat.shape[0]] = cat.to(cdev)
    gathered = [torch.empty_like(buf) for _ in range(w)]
    d.all_gather(gathered, buf)
    gathered = [g.to(device) for g in gathered]
    out, cursor = [None] * len(chunks), [0] * w
    for i in range(len(chunks)):
        k = owner(i, w)
        out[i] = gathered[k][cursor[k]:cursor[k] + lens[i]]
        cursor[k] += lens[i]
    return out


def all_gather_views(local, n_views, shape, dtype, device):
    """Per-view tensors computed by their owning rank (view i belongs to rank i % world, `local` in ascending view
    order) -> [n_views, *shape] on every rank with ONE all_gather (SURVEY.md §8e "map distribution")."""
    d = _dist()
    w, r = world(), rank()
    if not d:
        return torch.stack(local, 0) if local else torch.empty((0,) + tuple(shape), dtype=dtype, device=device)
    cap = (n_views + w - 1) // w
    cdev = _comm_device(device)
    buf = torch.zeros((cap,) + tuple(shape), dtype=dtype, device=cdev)
    for j, t in enumerate(local):
        buf[j] = t.to(cdev)
    # one flat receive buffer [w, cap, *shape] (all_gather_into_tensor: no per-rank tensor list, no flatten copy inside the
    # backend); view i sits at row (i % w) * cap + i // w
    gathered = torch.empty((w * cap,) + tuple(shape), dtype=dtype, device=cdev)      # (the concatenated form: both backends take it)
    d.all_gather_into_tensor(gathered, buf)
    idx = torch.arange(n_views, device=cdev)
    return gathered[(idx % w) * cap + idx // w].to(device)


def refine_sharded():
    """Does `refine` shard its per-point stages over the ranks?  Default since round 6: NO -- every rank runs the
    device-resident pass on all points (pmvo.py::_refine_device: ~10 ms at the headline size, no communication at all) and
    rank 0 writes the files.  The sharded smoothing loop pays two small all_gathers per 5000-point chunk -- 116 collectives of
    >= 20 us each at 58 chunks, i.e. at least what the WHOLE single-rank loop takes (3.6 ms) -- and hands its shell stage and
    voxel fit to the host-driven path.  MH_REFINE_SHARD=1 selects it (kept, and pinned to the reference's four-chunk run with 2
    and 3 ranks: tests/test_multichunk_gpu.py; tools/scale_first_run.sh measures both on the first multi-GPU node)."""
    import os

    d = _dist()
    if not d or world() == 1:
        return False
    env = os.environ.get("MH_REFINE_SHARD")
    return env == "1"


def all_gather_rows_inplace(buf, lo, s):
    """buf: device tensor [M] or [M, C] that is IDENTICAL on all ranks except for the rows each rank has just written,
    rows [lo + rank*s, lo + (rank+1)*s).  Afterwards every rank holds everybody's rows [lo, lo + world*s): one in-place
    all_gather (sendbuff = recvbuff + rank*count).  Rows of the range that nobody wrote travel too -- they are the same
    everywhere, so the copy is a no-op on them (buf must be allocated with world*s rows of slack at the end)."""
    d = _dist()
    R, r = world(), rank()
    out = buf[lo:lo + R * s]
    inp = buf[lo + r * s:lo + (r + 1) * s]
    assert out.shape[0] == R * s and out.is_contiguous(), "all_gather_rows_inplace: the buffer needs world*s rows of slack"
    if d.get_backend() == "nccl":
        if inplace_gather_supported(buf.device):
            d.all_gather_into_tensor(out, inp)
            return
        tmp = torch.empty_like(out)
        d.all_gather_into_tensor(tmp, inp.clone())
        out.copy_(tmp)
    else:
        host = inp.cpu()
        parts = [torch.empty_like(host) for _ in range(R)]
        d.all_gather(parts, host)
        out.copy_(torch.cat(parts, 0).to(buf.device))


_INPLACE_GATHER_OK = {}     # (device index, backend) -> (process group object, bool): decided ONCE per process group


def inplace_gather_supported(device):
    """Can this torch build's all_gather_into_tensor take an input that aliases its slice of the output (NCCL's in-place
    form, sendbuff == recvbuff + rank*count)?  Decided once, on a 256-byte probe, and AGREED by all ranks with one
    all_reduce(MIN): a rank that falls back on its own would issue a collective its peers never enter (deadlock).  An
    argument check that rejects the aliasing raises before any traffic (on every rank alike -- a probe that raised on some
    ranks only would leave the others inside the collective); the exception is logged, not swallowed.  The verdict is
    cached per (device, backend) TOGETHER WITH the process group object it was probed on (held, so its identity cannot be
    handed to a later group): a default group that was destroyed and made again is probed again; without a group object
    to compare with (a torch build without `group.WORLD`) nothing is cached."""
    import warnings

    d = _dist()
    pg = getattr(getattr(d, "group", None), "WORLD", None)
    key = (torch.device(device).index or 0, d.get_backend())
    hit = _INPLACE_GATHER_OK.get(key)
    if hit is None or pg is None or hit[0] is not pg:
        R, r = world(), rank()
        ok = 1
        try:
            probe = torch.full((R * 64,), float(r), dtype=torch.float32, device=device)
            d.all_gather_into_tensor(probe, probe[r * 64:(r + 1) * 64])
            want = torch.arange(R, dtype=torch.float32, device=device).repeat_interleave(64)
            if not torch.equal(probe, want):
                ok = 0
                warnings.warn("all_gather_into_tensor in place returned wrong rows on rank %d: using a separate receive "
                              "buffer" % r)
        except (RuntimeError, ValueError) as e:
            ok = 0
            warnings.warn("all_gather_into_tensor does not accept the in-place form here (%r): using a separate receive "
                          "buffer" % (e,))
        t = torch.tensor([ok], dtype=torch.int32, device=device)
        d.all_reduce(t, op=d.ReduceOp.MIN)
        hit = _INPLACE_GATHER_OK[key] = (pg, bool(int(t.item())))
    return hit[1]


_COMMS = {}
_CAPI_BROKEN = []          # non-empty once the hand-bound RCCL path failed to come up on some rank: stay on torch


def exchange_mode():
    """MH_VOLUME_EXCHANGE: torch (default) | capi | dense -- see the module docstring"""
    import os

    m = os.environ.get("MH_VOLUME_EXCHANGE", "torch").lower()
    if m not in ("torch", "capi", "dense"):
        raise ValueError("MH_VOLUME_EXCHANGE must be torch, capi or dense, not %r" % m)
    return m


def _destroy_comms():
    """ncclCommDestroy for every communicator made here (registered with atexit by rccl_comm)"""
    from . import _lib

    while _COMMS:
        _, (ctx, comm) = _COMMS.popitem()
        try:
            _lib.lib().mh_comm_destroy(comm)
        except Exception:
            pass


def rccl_comm(device):
    """The ncclComm_t of this job created through the C ABI (mh_comm_init): rank 0 draws the 128-byte unique id
    (mh_comm_unique_id), torch.distributed carries it to the other ranks, every rank joins.  Cached per device and
    destroyed at interpreter exit.  -> (ctx, comm) handles for mh_volume_reduce / mh_volume_gather."""
    import atexit
    import ctypes

    from . import _lib
    from .pmvo_utils import _ctx_for

    d = _dist()
    dev = torch.device(device)
    key = (dev.index or 0, world(), rank())
    if key not in _COMMS:
        L = _lib.lib()
        ctx = _ctx_for(dev)
        raw = (ctypes.c_ubyte * 128)()
        id_error = None
        if rank() == 0:
            try:
                _lib.check(L.mh_comm_unique_id(ctypes.cast(raw, ctypes.c_void_p)), "mh_comm_unique_id")
            except _lib.MhError as e:      # the peers are waiting in the broadcast below: tell them (an all-zero id)
                id_error = e
                raw = (ctypes.c_ubyte * 128)()
        if d and world() > 1:
            cdev = _comm_device(dev)
            t = torch.tensor(list(raw), dtype=torch.uint8, device=cdev)
            d.broadcast(t, src=0)
            raw = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        if id_error is not None or not any(raw):
            raise id_error or _lib.MhError("mh_comm_unique_id failed on rank 0")
        comm = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(L.mh_comm_init(ctx, ctypes.cast(raw, ctypes.c_void_p), world(), rank(), ctypes.byref(comm)),
                       "mh_comm_init")
        if not _COMMS:
            atexit.register(_destroy_comms)
        _COMMS[key] = (ctx, comm)
    return _COMMS[key]


def capi_comm_or_none(device):
    """rccl_comm, agreed on by ALL ranks: if the hand-bound communicator cannot be created on any rank (librccl missing,
    a binding error), every rank learns it through one torch.distributed all_reduce and the caller stays on the torch
    exchange -- a rank must never enter a collective its peers have given up on."""
    import warnings

    d = _dist()
    if _CAPI_BROKEN:
        return None
    ok, err = 1, None
    try:
        handles = rccl_comm(device)
    except Exception as e:          # MhError (library, symbol, ncclCommInitRank) -- reported below
        ok, err, handles = 0, e, None
    if d and world() > 1:
        t = torch.tensor([ok], dtype=torch.int32, device=_comm_device(device))
        d.all_reduce(t, op=d.ReduceOp.MIN)
        ok = int(t.item())
    if not ok:
        _CAPI_BROKEN.append(repr(err) if err else "a peer rank failed")
        warnings.warn("MH_VOLUME_EXCHANGE=capi: the C-ABI RCCL communicator did not come up (%s); using the "
                      "torch.distributed exchange" % _CAPI_BROKEN[0])
        return None
    return handles


def slab_bounds(grid_x, n_ranks):
    """x-slab ownership of the volume: rank r owns x in [b[r], b[r+1])"""
    return np.array([(int(grid_x) * r) // n_ranks for r in range(n_ranks + 1)], dtype=np.int32)


def volume_reduce(vol, device, mode=0, root=0):
    """mh_volume_reduce on a dense [X,Y,Z,C] fp32 device tensor whose x-slabs were filled by their owners: after the
    call (stream-ordered on the current stream) rank `root` holds the whole volume.  mode 0: slab gather
    (ncclSend/ncclRecv, direct to the root); mode 1: dense ncclReduce(sum)."""
    import ctypes

    from . import _lib

    assert vol.is_cuda and vol.dtype == torch.float32 and vol.is_contiguous() and vol.dim() == 4
    ctx, comm = rccl_comm(device)
    X, Y, Z, C = (int(v) for v in vol.shape)
    slabs = slab_bounds(X, world())
    with torch.cuda.device(vol.device):
        _lib.check(_lib.lib().mh_volume_reduce(ctx, comm, rank(), world(), root, _lib.ptr(vol), X, Y, Z, C,
                                               slabs.ctypes.data_as(ctypes.c_void_p), int(mode), _lib.stream_ptr()),
                   "mh_volume_reduce")
    return vol


def volume_gather(slab, vol, grid, device, root=0, handles=None):
    """mh_volume_gather: `slab` = this rank's own [b[r+1]-b[r], Y, Z, C] fp32 device tensor (on the root it may be the
    view vol[b[r]:b[r+1]] itself: no copy), `vol` = the dense [X,Y,Z,C] tensor on the root, None elsewhere."""
    import ctypes

    from . import _lib

    X, Y, Z, C = (int(v) for v in grid)
    slabs = slab_bounds(X, world())
    r = rank()
    assert slab is None or (slab.is_cuda and slab.dtype == torch.float32 and slab.is_contiguous()
                            and tuple(slab.shape) == (int(slabs[r + 1] - slabs[r]), Y, Z, C)), "slab shape"
    assert (vol is not None) == (r == root), "the dense volume lives on the root only"
    if vol is not None:
        assert vol.is_cuda and vol.dtype == torch.float32 and vol.is_contiguous() and tuple(vol.shape) == (X, Y, Z, C)
    ctx, comm = handles if handles is not None else rccl_comm(device)
    with torch.cuda.device(torch.device(device)):
        _lib.check(_lib.lib().mh_volume_gather(ctx, comm, r, world(), root, _lib.ptr(slab) if slab is not None and
                                               slab.numel() else None, _lib.ptr(vol), X, Y, Z, C,
                                               slabs.ctypes.data_as(ctypes.c_void_p), _lib.stream_ptr()),
                   "mh_volume_gather")
    return vol


def slab_gather_torch(slab, vol, grid, device, root=0):
    """The same exchange through torch.distributed's own binding: every peer isend's its slab, the root irecv's it in
    place (batch_isend_irecv = one ncclGroupStart/End under the nccl backend).  Under gloo (CPU tests, ranks sharing a
    GPU) the slabs are staged on the host."""
    d = _dist()
    w, r = world(), rank()
    X = int(grid[0])
    b = slab_bounds(X, w)
    on_gpu = d.get_backend() == "nccl"
    ops, staged = [], []
    if r == root:
        for k in range(w):
            if k == root or b[k + 1] == b[k]:
                continue
            dst = vol[b[k]:b[k + 1]]
            if not on_gpu:
                dst = torch.empty(dst.shape, dtype=dst.dtype, device="cpu")
                staged.append((k, dst))
            ops.append(d.P2POp(d.irecv, dst, k))
    elif slab is not None and slab.numel():
        ops.append(d.P2POp(d.isend, slab if on_gpu else slab.cpu(), root))
    if ops:
        for req in d.batch_isend_irecv(ops):
            req.wait()
    for k, t in staged:
        vol[b[k]:b[k + 1]] = t.to(vol.device)
    return vol


def voxel_owner_mask(x, n_ranks, r, grid_x):
    """Spatial partition of the volume into n_ranks slabs along x: rank r owns x in [r*G/n, (r+1)*G/n)."""
    lo = (grid_x * r) // n_ranks
    hi = (grid_x * (r + 1)) // n_ranks
    return (x >= lo) & (x < hi)


def voxel_fit_reduced(select_points, select_ori, device, voxel_min, voxel_size, grid_resolution, fit=None,
                      sparse=False):
    """Voxel fit with disjoint voxel ownership + the single exchange.  Returns dense (occ [X,Y,Z], ori [X,Y,Z,3])
    float64 numpy arrays on rank 0 (zeros elsewhere); with sparse=True the occupied voxels instead:
    (voxels [G,3] int64 (x,y,z) in ascending voxel order, ori [G,3] float32), empty off rank 0."""
    from . import pmvo_utils as U

    fit = U.voxel_fit if fit is None else fit
    g = np.asarray(grid_resolution).astype(np.int64)
    d = _dist()
    w, r = world(), rank()
    pts, ori = select_points, select_ori          # (the fit does not modify its inputs)
    if fit is not U.voxel_fit:                    # a caller-supplied fit may, like the reference's in-place flips
        pts, ori = np.array(select_points, copy=True), np.array(select_ori, copy=True)
    if not d:
        res = fit(pts, ori, device, voxel_min, voxel_size, g, dense=not sparse)
        if sparse:
            return res["voxels"].cpu().numpy(), res["ori"].cpu().numpy()
        return res["occ"], res["ori_dense"]
    probe = pts.copy()
    x, _, _ = U.p2v(probe, np.asarray(voxel_min), voxel_size, g)
    own = voxel_owner_mask(x, w, r, int(g[0]))
    X, Y, Z = int(g[0]), int(g[1]), int(g[2])
    b = slab_bounds(X, w)
    lo, hi = int(b[r]), int(b[r + 1])
    mode = exchange_mode()
    handles = capi_comm_or_none(device) if (mode == "capi" and torch.device(device).type == "cuda") else None
    if mode == "capi" and handles is None:
        mode = "torch"
    # peers hold their own slab only; the root holds the shared volume (the dense-reduce comparison mode needs it everywhere)
    if r == 0 or mode == "dense":
        vol = torch.zeros((X, Y, Z, 4), dtype=torch.float32, device=device)
        slab = vol[lo:hi]
    else:
        vol = None
        slab = torch.zeros((hi - lo, Y, Z, 4), dtype=torch.float32, device=device)
    if own.any():
        res = fit(pts[own], ori[own], device, voxel_min, voxel_size, g, dense=False)
        v = res["voxels"].to(device)
        slab[v[:, 0] - lo, v[:, 1], v[:, 2], 0] = 1.0
        slab[v[:, 0] - lo, v[:, 1], v[:, 2], 1:] = res["ori"].to(device)
    # the one exchange of the data path
    if mode == "capi":
        volume_gather(slab, vol, (X, Y, Z, 4), device, root=0, handles=handles)
    elif mode == "dense":
        if d.get_backend() == "nccl":
            d.reduce(vol, dst=0, op=d.ReduceOp.SUM)
        else:
            host = vol.to("cpu")
            d.reduce(host, dst=0, op=d.ReduceOp.SUM)
            vol = host.to(device) if r == 0 else vol
    else:
        slab_gather_torch(slab, vol, (X, Y, Z, 4), device, root=0)
    if sparse:
        if r != 0:
            return np.zeros((0, 3), np.int64), np.zeros((0, 3), np.float32)
        nz = torch.nonzero(vol[..., 0])                     # row-major order == ascending voxel key
        return nz.cpu().numpy(), vol[nz[:, 0], nz[:, 1], nz[:, 2], 1:].cpu().numpy()
    if r != 0:
        return np.zeros(tuple(g)), np.zeros(tuple(g) + (3,))
    vol = vol.cpu().numpy()
    return vol[..., 0].astype(np.float64), vol[..., 1:].astype(np.float64)
