"""Multi-GPU plumbing of the PMVO path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by POINTS, not by views: a point's result needs every view but no other point, every GPU
holds all packed views (2.5 GB at 60 x 1080p, nothing against 288 GB), so an iteration has no collective at
all and results are bit-identical to the single-GPU run.  Exactly two kinds of exchange exist:
  * map_chunks: independent chunks dealt round-robin, one all_gather of the per-chunk results at the end;
  * voxel_fit_reduced: every rank fits a disjoint x-slab of voxels and writes them into a zero volume; ONE
    exchange over xGMI assembles the shared 3D orientation/occupancy volume on rank 0: mh_volume_reduce (RCCL
    through the C ABI) -- because ownership is disjoint, every peer sends its slab straight to the root
    (ncclSend/ncclRecv), 1/N of the bytes of a dense reduce per link; the dense ncclReduce(sum) is kept as mode 1
    (x + 0 is exact, so both give the single-GPU volume bit for bit).
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


def map_chunks(chunks, fn, device, empty, after=None):
    """Apply fn to every chunk (each rank takes chunks i with i % world == rank) and return the list of all
    results in chunk order on every rank.  fn returns a 2-D tensor on `device`; `empty()` gives a [0,C] one.
    after(): called once all chunks have been issued and before any result is read (e.g. to join side streams)."""
    d = _dist()
    w, r = world(), rank()
    mine = {i: fn(c) if len(c) else empty() for i, c in enumerate(chunks) if owner(i, w) == r}
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
    gathered = [torch.empty_like(buf) for _ in range(w)]
    d.all_gather(gathered, buf)
    out = torch.empty((n_views,) + tuple(shape), dtype=dtype, device=device)
    for i in range(n_views):
        out[i] = gathered[owner(i, w)][i // w].to(device)
    return out


_COMMS = {}


def rccl_comm(device):
    """The ncclComm_t of this job created through the C ABI (mh_comm_init): rank 0 draws the 128-byte unique id
    (mh_comm_unique_id), torch.distributed carries it to the other ranks, every rank joins.  Cached per device.
    -> (ctx, comm) handles for mh_volume_reduce."""
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
        if rank() == 0:
            _lib.check(L.mh_comm_unique_id(ctypes.cast(raw, ctypes.c_void_p)), "mh_comm_unique_id")
        if d and world() > 1:
            cdev = _comm_device(dev)
            t = torch.tensor(list(raw), dtype=torch.uint8, device=cdev)
            d.broadcast(t, src=0)
            raw = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        comm = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(L.mh_comm_init(ctx, ctypes.cast(raw, ctypes.c_void_p), world(), rank(), ctypes.byref(comm)),
                       "mh_comm_init")
        _COMMS[key] = (ctx, comm)
    return _COMMS[key]


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


def voxel_owner_mask(x, n_ranks, r, grid_x):
    """Spatial partition of the volume into n_ranks slabs along x: rank r owns x in [r*G/n, (r+1)*G/n)."""
    lo = (grid_x * r) // n_ranks
    hi = (grid_x * (r + 1)) // n_ranks
    return (x >= lo) & (x < hi)


def voxel_fit_reduced(select_points, select_ori, device, voxel_min, voxel_size, grid_resolution, fit=None,
                      sparse=False):
    """Voxel fit with disjoint voxel ownership + the single reduce.  Returns dense (occ [X,Y,Z], ori [X,Y,Z,3])
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
    vol = torch.zeros((int(g[0]), int(g[1]), int(g[2]), 4), dtype=torch.float32, device=device)
    if own.any():
        res = fit(pts[own], ori[own], device, voxel_min, voxel_size, g, dense=False)
        v = res["voxels"].to(device)
        vol[v[:, 0], v[:, 1], v[:, 2], 0] = 1.0
        vol[v[:, 0], v[:, 1], v[:, 2], 1:] = res["ori"].to(device)
    # the one exchange of the data path: RCCL through the C ABI (slab gather over xGMI, direct to rank 0); under the
    # gloo backend of the CPU / single-GPU multi-rank tests the same volume is summed by torch.distributed on the host
    if d.get_backend() == "nccl":
        volume_reduce(vol, device, mode=0, root=0)
    else:
        vol = vol.to("cpu")
        d.reduce(vol, dst=0, op=d.ReduceOp.SUM)
    if sparse:
        if r != 0:
            return np.zeros((0, 3), np.int64), np.zeros((0, 3), np.float32)
        vol = vol.to(device)
        nz = torch.nonzero(vol[..., 0])                     # row-major order == ascending voxel key
        return nz.cpu().numpy(), vol[nz[:, 0], nz[:, 1], nz[:, 2], 1:].cpu().numpy()
    if r != 0:
        return np.zeros(tuple(g)), np.zeros(tuple(g) + (3,))
    vol = vol.cpu().numpy()
    return vol[..., 0].astype(np.float64), vol[..., 1:].astype(np.float64)
