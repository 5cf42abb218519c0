"""oracle -- TEST INFRASTRUCTURE ONLY (CPU restatement of the reference's PMVO path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (monohair_amd) never does.  See oracle/pmvo_oracle.c for the
per-function reference citations and the parity status (PINNED against golden vectors
generated from the imported reference by tools/gen_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
CAM_STRIDE = 48

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int32)
c_u8 = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def _p(a, t=c_f):
    return None if a is None else a.ctypes.data_as(t)


class _Views(ctypes.Structure):
    _fields_ = [
        ("V", ctypes.c_int),
        ("H", ctypes.c_int),
        ("W", ctypes.c_int),
        ("cams", c_f),
        ("depth", ctypes.POINTER(c_f)),
        ("ori", ctypes.POINTER(c_f)),
        ("conf", ctypes.POINTER(c_f)),
        ("mask", ctypes.POINTER(c_f)),
    ]


class Views:
    """Compact per-view planes + camera records, as the C oracle wants them.

    depth/conf/mask: [V,H,W] float32, ori: [V,H,W,2] float32 (channel 0 of the reference's
    3-channel depth/mask arrays), cams: [V,48] float32 records (monohair_amd.camera.Camera.record)."""

    def __init__(self, cams, depth, ori, conf, mask):
        self.cams = np.ascontiguousarray(cams, dtype=np.float32)
        self.depth = np.ascontiguousarray(depth, dtype=np.float32)
        self.ori = np.ascontiguousarray(ori, dtype=np.float32)
        self.conf = np.ascontiguousarray(conf, dtype=np.float32)
        self.mask = np.ascontiguousarray(mask, dtype=np.float32)
        self.V, self.H, self.W = self.depth.shape
        assert self.cams.shape == (self.V, CAM_STRIDE)
        arr = c_f * self.V
        self._pd = arr(*[_p(self.depth[v]) for v in range(self.V)])
        self._po = arr(*[_p(self.ori[v]) for v in range(self.V)])
        self._pc = arr(*[_p(self.conf[v]) for v in range(self.V)])
        self._pm = arr(*[_p(self.mask[v]) for v in range(self.V)])
        self.c = _Views(self.V, self.H, self.W, _p(self.cams), self._pd, self._po, self._pc, self._pm)

    @classmethod
    def from_reference_dicts(cls, cameras, depths, Ori, Conf, masks):
        """Same argument shapes as the reference's PMVO.__init__ (PMVO.py:14-28)."""
        from monohair_amd.camera import camera_records

        keys = list(cameras.keys())
        d = np.stack([np.asarray(depths[k], dtype=np.float32)[..., 0] for k in keys])
        o = np.stack([np.asarray(Ori[k]).astype(np.float32) for k in keys])
        c = np.stack([np.asarray(Conf[k]).astype(np.float32) for k in keys])
        m = np.stack([np.asarray(masks[k]).astype(np.float32)[..., 0] for k in keys])
        return cls(camera_records(cameras), d, o, c, m)


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def num_threads():
    return int(lib().orc_num_threads())


def project_points(cam_rec, pts, H, W):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    N = pts.shape[0]
    rc = np.empty((N, 2), np.int32)
    zp = np.empty(N, np.float32)
    oob = np.empty(N, np.uint8)
    pixf = np.empty((N, 2), np.float32)
    rec = np.ascontiguousarray(cam_rec, dtype=np.float32)
    lib().orc_project_points(_p(rec), _p(pts), N, H, W, _p(rc, c_i), _p(zp), _p(oob, c_u8), _p(pixf))
    return rc, zp, oob.astype(bool), pixf


def _side(patch):
    """the reference's tap window for a given patch_size: range(-(p // 2), p // 2 + 1) (PMVO.py:494-495), i.e. an even
    size samples the next odd window"""
    return 2 * (int(patch) // 2) + 1


def visible_and_ori(views, pts, patch):
    patch = _side(patch)
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    N, V, P = pts.shape[0], views.V, patch * patch
    out = dict(
        visible=np.empty((V, N), np.float32),
        Ori=np.empty((V, N, 2), np.float32),
        Conf=np.empty((V, N), np.float32),
        mask=np.empty((V, N), np.float32),
        Ori_patch=np.empty((V, N, P, 2), np.float32),
        Conf_patch=np.empty((V, N, P), np.float32),
        pixf=np.empty((V, N, 2), np.float32),
    )
    lib().orc_visible_and_ori(
        ctypes.byref(views.c), _p(pts), N, patch, _p(out["visible"]), _p(out["Ori"]), _p(out["Conf"]),
        _p(out["mask"]), _p(out["Ori_patch"]), _p(out["Conf_patch"]), _p(out["pixf"]))
    return out


def topk_views(vis, conf, k=20, order="torch"):
    """Find_max_conf_from_visible_view.  order "torch": equal values in torch.topk's CPU order (std::nth_element +
    std::sort, topk_oracle.cpp); "index": value descending, view index ascending (the library's topk_order = 1)."""
    vis = np.ascontiguousarray(vis, np.float32)
    conf = np.ascontiguousarray(conf, np.float32)
    V, N = vis.shape
    idx = np.empty((k, N), np.int32)
    val = np.empty((k, N), np.float32)
    fn = lib().orc_topk_views if order == "torch" else lib().orc_topk_views_by_index
    fn(_p(vis), _p(conf), V, N, k, _p(idx, c_i), _p(val))
    return idx, val


def topk_column(values, k):
    """torch.topk(values, k) of one float32 vector -> (indices, values) in the CPU kernel's order"""
    v = np.ascontiguousarray(values, np.float32)
    idx = np.empty(k, np.int32)
    val = np.empty(k, np.float32)
    lib().orc_topk_column(_p(v), len(v), k, _p(idx, c_i), _p(val))
    return idx, val


REPROJECT_RULES = {"group": 0, "mid": 1, "chain": 2}
REF_FMA_MIN_COLS = 28445   # MKL 2024.2 / AVX-512 / 8 threads: the container the goldens were generated in


def set_reproject_rule(mode="group", fma_min_cols=REF_FMA_MIN_COLS):
    """How sample_next_3d_pos's sgemms round (oracle/pmvo_oracle.c, the table above cam_unproject): "group" follows the
    number of points that share a (rank, base view) in the batch, as the reference's MKL calls do; "mid" / "chain" force
    one form for every point.  Returns the previous (mode, fma_min_cols)."""
    prev = get_reproject_rule()
    lib().orc_set_reproject_rule(REPROJECT_RULES[mode], ctypes.c_longlong(int(fma_min_cols)))
    return prev


def get_reproject_rule():
    m, c = ctypes.c_int(0), ctypes.c_longlong(0)
    lib().orc_get_reproject_rule(ctypes.byref(m), ctypes.byref(c))
    return {v: k for k, v in REPROJECT_RULES.items()}[m.value], c.value


def set_sum_block(cols=32):
    """Columns per vectorised block of ATen's outer sum (oracle/pmvo_oracle.c: row_sum1): the trailing C mod cols columns
    of a [V, C] sum(dim=0) are added in another order.  32 where the goldens were generated (ATen dispatches this kernel at AVX2 width); 0 = cascade
    everywhere.  Returns the previous value."""
    prev = lib().orc_get_sum_block()
    lib().orc_set_sum_block(int(cols))
    return prev


def mm4(A, B):
    """[4,4] @ [4,C] as the reference's torch.matmul rounds it under the rule in force (oracle/pmvo_oracle.c: orc_mm4)"""
    A, B = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(B, np.float32)
    out = np.empty_like(B)
    lib().orc_mm4(_p(A), _p(B), ctypes.c_longlong(B.shape[1]), _p(out))
    return out


def mm3(A, B):
    """[3,3] @ [3,C] likewise (Camera.reprojection's product, Utils/Camera_utils.py:103)"""
    A, B = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(B, np.float32)
    out = np.empty_like(B)
    lib().orc_mm3(_p(A), _p(B), ctypes.c_longlong(B.shape[1]), _p(out))
    return out


def outer_sum(x):
    """torch.sum(x, dim=0) of a contiguous [V, C] float32 array in ATen's order (cascade; row_sum for the trailing columns)"""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape[1], np.float32)
    lib().orc_outer_sum(_p(x), x.shape[0], ctypes.c_longlong(x.shape[1]), _p(out))
    return out


def group_sizes(base_view, V):
    base_view = np.ascontiguousarray(base_view, np.int32)
    cnt = np.empty(V, np.int32)
    lib().orc_group_sizes(_p(base_view, c_i), base_view.shape[0], V, _p(cnt, c_i))
    return cnt


def sample_next(views, pts, base_view, ori_c, offsets):
    pts = np.ascontiguousarray(pts, np.float32)
    base_view = np.ascontiguousarray(base_view, np.int32)
    ori_c = np.ascontiguousarray(ori_c, np.float32)
    offsets = np.ascontiguousarray(offsets, np.float32)
    N, S = pts.shape[0], offsets.shape[0]
    out = np.empty((N, S, 3), np.float32)
    lib().orc_sample_next(ctypes.byref(views.c), _p(pts), N, _p(base_view, c_i), _p(ori_c), _p(offsets), S, _p(out))
    return out


def reproject_ori(views, pts, samples):
    pts = np.ascontiguousarray(pts, np.float32)
    samples = np.ascontiguousarray(samples, np.float32)
    N, S = samples.shape[0], samples.shape[1]
    D = np.empty((views.V, N, S, 2), np.float32)
    lib().orc_reproject_ori(ctypes.byref(views.c), _p(pts), _p(samples), N, S, _p(D))
    return D


def prj_loss(D, ori_patch, conf_patch, vis, thr, want_all=False):
    D = np.ascontiguousarray(D, np.float32)
    ori_patch = np.ascontiguousarray(ori_patch, np.float32)
    conf_patch = np.ascontiguousarray(conf_patch, np.float32)
    vis = np.ascontiguousarray(vis, np.float32)
    V, N, S, _ = D.shape
    P = conf_patch.shape[-1]
    loss = np.empty(N, np.float32)
    idx = np.empty(N, np.int32)
    hc = np.empty(N, np.uint8)
    all_ = np.empty((N, S), np.float32) if want_all else None
    lib().orc_prj_loss(V, N, S, P, ctypes.c_float(thr), _p(D), _p(ori_patch), _p(conf_patch), _p(vis), _p(loss),
                       _p(idx, c_i), _p(hc, c_u8), _p(all_))
    if want_all:
        return loss, idx, hc.astype(bool), all_
    return loss, idx, hc.astype(bool)


def forward(views, pts, patch, thr, offsets, base_idx=None, base_val=None, nrank=10, rank_step=2, extra=False):
    """PMVO.forward (PMVO.py:39-78): returns (points, line_ori, min_loss, high_conf[, extras])."""
    patch = _side(patch)
    pts = np.ascontiguousarray(pts, np.float32)
    offsets = np.ascontiguousarray(offsets, np.float32)
    N, S = pts.shape[0], offsets.shape[0]
    if base_idx is not None:
        base_idx = np.ascontiguousarray(base_idx, np.int32)
        base_val = np.ascontiguousarray(base_val, np.float32)
    lo = np.empty((N, 3), np.float32)
    ml = np.empty(N, np.float32)
    hc = np.empty(N, np.uint8)
    bs = np.empty((N, 3), np.float32)
    br = np.empty(N, np.int32)
    bi = np.empty(N, np.int32)
    lib().orc_forward(ctypes.byref(views.c), _p(pts), N, patch, ctypes.c_float(thr), _p(offsets), S, nrank,
                      rank_step, _p(base_idx, c_i), _p(base_val), _p(lo), _p(ml), _p(hc, c_u8), _p(bs),
                      _p(br, c_i), _p(bi, c_i))
    if extra:
        return pts, lo, ml, hc.astype(bool), dict(best_sample=bs, best_rank=br, best_s=bi)
    return pts, lo, ml, hc.astype(bool)


def refine_loss(views, pts, dirs, patch, thr, mul=0.005, div=4.0):
    """Loss of one given direction per point: the core of PMVO.refine (PMVO.py:86-90)."""
    patch = _side(patch)
    pts = np.ascontiguousarray(pts, np.float32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    N = pts.shape[0]
    loss = np.empty(N, np.float32)
    hc = np.empty(N, np.uint8)
    lib().orc_refine_loss(ctypes.byref(views.c), _p(pts), _p(dirs), ctypes.c_float(mul), ctypes.c_float(div), N,
                          patch, ctypes.c_float(thr), _p(loss), _p(hc, c_u8))
    return loss, hc.astype(bool)


def filter_votes(views, pts, patch, thr, vis_thr):
    """(surface_index, filter_index, unvisible_index, head_filter_votes) -- PMVO.py:402-480, :110-137."""
    patch = _side(patch)
    pts = np.ascontiguousarray(pts, np.float32)
    N = pts.shape[0]
    outs = [np.empty(N, np.uint8) for _ in range(4)]
    lib().orc_filter_points(ctypes.byref(views.c), _p(pts), N, patch, ctypes.c_float(thr), ctypes.c_float(vis_thr),
                            *[_p(o, c_u8) for o in outs])
    return [o.astype(bool) for o in outs]


def medoid_dense(ori):
    """compute_points_similarity (PMVO_utils.py:366-382): ori [G,K,3] -> (medoid [G,3], index [G])."""
    ori = np.ascontiguousarray(ori, np.float32)
    G, K, _ = ori.shape
    out = np.empty((G, 3), np.float32)
    idx = np.empty(G, np.int32)
    lib().orc_medoid_dense(_p(ori), G, K, _p(out), _p(idx, c_i))
    return out, idx


def medoid_segmented(ori, seg_start):
    ori = np.ascontiguousarray(ori, np.float32)
    seg_start = np.ascontiguousarray(seg_start, np.int32)
    G = len(seg_start) - 1
    out = np.zeros((G, 3), np.float32)
    idx = np.zeros(G, np.int32)
    lib().orc_medoid_segmented(_p(ori), _p(seg_start, c_i), G, _p(out), _p(idx, c_i))
    return out, idx


def replace_dissimilar(center, ori, thr=0.95):
    """refine's replacement rule (PMVO.py:631-636), in place on `ori` [N,3] float32; returns the number of replaced rows."""
    center = np.ascontiguousarray(center, np.float32)
    assert ori.dtype == np.float32 and ori.flags.c_contiguous and ori.shape == center.shape
    return int(lib().orc_replace_dissimilar(_p(center), _p(ori), ctypes.c_float(thr), ori.shape[0]))


def head_top_mask(points, scalp_tree, scalp_max):
    """The scalp half of filter_head_points (PMVO.py:112-121): within 4 cm of the scalp and below its top by 1 cm."""
    pts = np.asarray(points)
    d, _ = scalp_tree.query(pts, k=1)
    return np.logical_and(d < 0.04, pts[:, 2] < scalp_max[2] - 0.01)


def refine_loop(views, points, ori, loss, patch, thr, vis_thr, scalp_tree, scalp_max, sub_num=5000, k=100, sim_thr=0.95,
                workers=-1, trace=None):
    """The smoothing loop of refine (PMVO.py:602-643), IN PLACE on `ori` [N,3] / `loss` [N] (float32) like the reference:
    chunks of `sub_num` points in order; a chunk's neighbour orientations are read from `ori` AS IT IS when the chunk starts
    (:612 -- earlier chunks have already written back, :640), medoid of the k nearest (self included, scipy's order),
    loss of that direction (PMVO.refine, :82-93: head-filtered points get -1), replacement where |cos| < 0.95 (:631-636),
    -1 -> 0.5 (:639), write-back (:640-641).  `step = N // sub_num + 1` (:603): when N is a multiple of sub_num the
    reference enters one more chunk with zero points; see tests/golden/e2e_multichunk.npz (`exact_raised`) for what it does
    there -- this restatement skips it.
    points: [N,3] as the reference holds them (the float32 select_p.npy); the KDTree is built on them as given.
    trace: optional list that receives (lo, hi, replaced) per chunk."""
    from scipy.spatial import KDTree

    assert ori.dtype == np.float32 and loss.dtype == np.float32
    N = points.shape[0]
    tree = KDTree(data=points)
    kk = min(k, N)
    step = N // sub_num + 1
    for i in range(step):
        lo, hi = i * sub_num, min((i + 1) * sub_num, N)
        if hi <= lo:
            continue
        sub_points = np.ascontiguousarray(points[lo:hi], np.float32)
        _, index = tree.query(points[lo:hi], kk, workers=workers)
        index = np.asarray(index).reshape(hi - lo, kk)
        center, _ = medoid_dense(ori[index])                              # ori as of NOW
        update_loss, _ = refine_loss(views, sub_points, center, patch, thr)
        votes = filter_votes(views, sub_points, patch, thr, vis_thr)[3]
        filt = votes & ~head_top_mask(sub_points, scalp_tree, scalp_max)  # filter_head_points (:110-137)
        update_loss[filt] = -1
        sub_ori = ori[lo:hi].copy()
        replaced = replace_dissimilar(center, sub_ori, sim_thr)
        update_loss[update_loss == -1] = 0.5
        ori[lo:hi] = sub_ori
        loss[lo:hi] = update_loss
        if trace is not None:
            trace.append((lo, hi, replaced))
    return ori, loss


def shell_orientations(views, select_points, select_ori, shell_points, patch, thr, vis_thr, scalp_tree, scalp_max, k=100,
                       workers=-1):
    """Orientations for the occluded shell points (PMVO.py:655-691): medoid of the k nearest KEPT surface points (scipy's
    order), minus the points the head filter removes.  The reference walks the shell points in chunks of 5000; they are
    independent of each other, so one pass gives the same rows.  -> (kept shell points [M,3] float32, orientations [M,3])."""
    from scipy.spatial import KDTree

    tree = KDTree(data=select_points)
    _, index = tree.query(shell_points, min(k, len(select_points)), workers=workers)
    centre, _ = medoid_dense(np.asarray(select_ori, np.float32)[np.asarray(index).reshape(len(shell_points), -1)])
    sub = np.ascontiguousarray(shell_points, np.float32)
    votes = filter_votes(views, sub, patch, thr, vis_thr)[3]
    filt = votes & ~head_top_mask(sub, scalp_tree, scalp_max)
    return sub[~filt], centre[~filt]


def p2v(points, voxel_min, voxel_size, grid_resolution):
    """p2v (PMVO_utils.py:386-404): flips y,z IN PLACE, float64 round-half-even, clip."""
    points[:, 1:] *= -1
    idx = np.round((points - voxel_min) / voxel_size).astype(np.int32)
    g = np.asarray(grid_resolution)
    idx = np.clip(idx, 0, g - 1)
    return idx[:, 0], idx[:, 1], idx[:, 2]


def voxel_fit(select_points, select_ori, voxel_min, voxel_size, grid_resolution):
    """The voxel fit of refine (PMVO.py:695-726): sign canonicalisation, p2v, groups in first-occurrence /
    point order, per-voxel medoid.  Returns (occ [X,Y,Z] f64, ori [X,Y,Z,3] f64); mutates its inputs like the
    reference does."""
    g = np.asarray(grid_resolution).astype(np.int64)
    occ = np.zeros(tuple(g))
    ori = np.zeros(tuple(g) + (3,))
    up = select_ori[:, 1] > 0
    select_ori[up] *= -1
    x, y, z = p2v(select_points, np.asarray(voxel_min), voxel_size, g)
    key = (x.astype(np.int64) * g[1] + y) * g[2] + z
    order = np.argsort(key, kind="stable")          # stable: point order inside every voxel is kept
    ks = key[order]
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
    seg = np.concatenate([starts, [len(ks)]]).astype(np.int32)
    med, _ = medoid_segmented(select_ori[order].astype(np.float32), seg)
    vx, vy, vz = x[order][starts], y[order][starts], z[order][starts]
    occ[vx, vy, vz] = 1
    ori[vx, vy, vz] = med.astype(np.float64)
    return occ, ori


def mat_layout(occ, ori):
    """The on-disk layout of Ori3D.mat / Occ3D.mat (PMVO.py:753-756): ori [X,Y,Z,3] -> [Y,X,3*Z] with last index
    c*Z+z, occ [X,Y,Z] -> [Y,X,Z]."""
    g = occ.shape
    o = ori.transpose((0, 1, 3, 2)).reshape(g[0], g[1], g[2] * 3).transpose((1, 0, 2))
    return o, occ.transpose((1, 0, 2))


def gabor_bank(bank, img, want_sum=False):
    """calOrientationGabor.filter/forward, iter=1 (GaborFilter.py:29-113): (orient index, conf, variance[, the sum under the
    variance's root])."""
    bank = np.ascontiguousarray(bank, np.float32).reshape(180, 17, 17)
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape
    orient = np.empty((H, W), np.int32)
    conf = np.empty((H, W), np.float32)
    var = np.empty((H, W), np.float32)
    if want_sum:
        vs = np.empty((H, W), np.float32)
        lib().orc_gabor_bank_sums(_p(bank), _p(img), H, W, _p(orient, c_i), _p(conf), _p(var), _p(vs))
        return orient, conf, var, vs
    lib().orc_gabor_bank(_p(bank), _p(img), H, W, _p(orient, c_i), _p(conf), _p(var))
    return orient, conf, var


# ------------------------------------------------------------------------------------------- strand tracing
class _Volume(ctypes.Structure):
    _fields_ = [("W", ctypes.c_int), ("H", ctypes.c_int), ("Z", ctypes.c_int), ("vox", c_f)]


class Volume:
    """vox[Z,H,W,4] = (ori_x, ori_y, ori_z, occ) as HairGrowing.__init__ holds it (HairGrow.py:41-55: ori[1:] negated)."""

    def __init__(self, occ_zyx, ori_zyx3):
        occ = np.asarray(occ_zyx, np.float32)
        ori = np.asarray(ori_zyx3, np.float32).copy()
        ori[..., 1:] *= -1
        self.vox = np.ascontiguousarray(np.concatenate([ori, occ[..., None]], -1), np.float32)
        self.Z, self.H, self.W = occ.shape
        self.c = _Volume(self.W, self.H, self.Z, _p(self.vox))


def trace_seeds(vol, seeds, thr):
    seeds = np.ascontiguousarray(seeds, np.float32)
    n = len(seeds)
    out = np.zeros((n, 513, 3), np.float32)
    first = np.zeros(n, np.int32)
    ln = np.zeros(n, np.int32)
    lib().orc_trace_seeds(ctypes.byref(vol.c), _p(seeds), n, ctypes.c_float(thr), _p(out), _p(first, c_i), _p(ln, c_i))
    return out, first, ln


def trace_scalp(vol, seeds, normals, thr):
    seeds = np.ascontiguousarray(seeds, np.float32)
    normals = np.ascontiguousarray(normals, np.float32)
    n = len(seeds)
    out = np.zeros((n, 257, 3), np.float32)
    ln = np.zeros(n, np.int32)
    lib().orc_trace_scalp(ctypes.byref(vol.c), _p(seeds), _p(normals), n, ctypes.c_float(thr), _p(out), _p(ln, c_i))
    return out, ln


def accept_strands(vol, flag, pts, first, ln, seeds, mode):
    n, stride = pts.shape[0], pts.shape[1]
    acc = np.zeros(n, np.uint8)
    seeds = np.ascontiguousarray(seeds, np.float32)
    lib().orc_accept_strands(vol.W, vol.H, vol.Z, _p(flag), _p(pts), _p(first, c_i), _p(ln, c_i), stride, _p(seeds), n,
                             mode, _p(acc, c_u8))
    return acc.astype(bool)


def voxel_seed_rounds(vol, flag, thr, jitter, rounds):
    """`rounds` passes of trace() over the occupied voxels (HairGrow.py:254-262 / :286-294).  The seed tensor is
    shifted IN PLACE by every call (:62-63: += 0.5, += rand*0.5), so the shifts accumulate from round to round.
    jitter: [rounds*n_occ, 3] uniform draws in the order the reference consumes them.  Mutates flag."""
    occ_idx = np.argwhere(vol.vox[..., 3] != 0)[:, ::-1].astype(np.float32)      # (x,y,z), z-major order (torch.nonzero)
    n = len(occ_idx)
    seeds_all, pos = [], occ_idx.copy()
    for rnd in range(rounds):
        pos = (pos + np.float32(0.5)).astype(np.float32)
        pos = (pos + (jitter[rnd * n:(rnd + 1) * n].astype(np.float32) * np.float32(0.5)).astype(np.float32)).astype(
            np.float32)
        seeds_all.append(pos.copy())
    seeds_all = np.concatenate(seeds_all, 0)
    tp, tf, tl = trace_seeds(vol, seeds_all, thr)
    acc = accept_strands(vol, flag, tp, tf, tl, seeds_all, 0)
    return [tp[i, tf[i]:tf[i] + tl[i]].copy() for i in np.flatnonzero(acc)]


def generate_guide_strands(vol, scalp_points, scalp_normals, thr, jitter):
    """GenerateGuideStrandFromScalp (HairGrow.py:226-265): roots traced from the scalp, then two rounds of
    voxel-seeded tracing gated by the flag volume.  Returns (strands in voxel space, num_root, flag)."""
    flag = np.zeros((vol.Z, vol.H, vol.W), np.float32)
    sp, sl = trace_scalp(vol, scalp_points, scalp_normals, thr)
    acc = accept_strands(vol, flag, sp, np.zeros(len(sl), np.int32), sl, scalp_points, 1)
    strands = [sp[i, :sl[i]].copy() for i in np.flatnonzero(acc)]
    num_root = len(strands)
    strands += voxel_seed_rounds(vol, flag, thr, jitter, 2)
    return strands, num_root, flag


def randomly_generate_segments(vol, thr, jitter):
    """randomlyGenerateSegments (HairGrow.py:269-299): three rounds of voxel-seeded tracing."""
    flag = np.zeros((vol.Z, vol.H, vol.W), np.float32)
    return voxel_seed_rounds(vol, flag, thr, jitter, 3), flag


def set_subpixel_bits(bits):
    """Sub-pixel grid of the two rasteriser statements: window positions are snapped to 2^-bits pixel (8 = shipped, 4 =
    what Google SwiftShader uses; the ctx option "raster_subpixel_bits" of the HIP library).  Stays set until changed."""
    lib().ora_set_subpixel_bits(int(bits))


def render_depth(cam_rec, verts, faces, H, W, pixel_center=0.5, channels=1):
    """CPU statement of monohair_amd/csrc/raster.hip (oracle/raster_oracle.c) -> (depth [H,W(,channels)], covered)."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    rec = np.ascontiguousarray(cam_rec, dtype=np.float32)
    out = np.empty((H, W, channels), np.float32)
    L = lib()
    L.ora_render_depth.restype = ctypes.c_long
    L.ora_render_depth.argtypes = [c_f, c_f, ctypes.c_int, c_i, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_float, c_f, ctypes.c_int]
    n = L.ora_render_depth(_p(rec), _p(verts), len(verts), _p(faces, c_i), len(faces), H, W, pixel_center, _p(out),
                           channels)
    if n < 0:
        raise MemoryError("ora_render_depth")
    return (out[..., 0] if channels == 1 else out), int(n)


def render_strands(cam_rec, verts, faces, line_pts, line_tan, H, W, pixel_center=0.5, width=3, color_option=2,
                   depth_option=1, clear=0.0, line_rule=0):
    """CPU statement of mh_render_strands (oracle/raster_oracle.c: ora_render_strands) -> (rgb [H,W,3] float32,
    prim [H,W] int32 (-1 background, < len(faces) mesh, else len(faces)+segment), pixels owned by strands).
    line_rule 0: GL's diamond-exit rule; 1: the end pixel of a segment is drawn too (the ctx option "line_rule")."""
    verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
    lp = np.ascontiguousarray(line_pts, dtype=np.float32).reshape(-1, 3)
    lt = np.ascontiguousarray(line_tan, dtype=np.float32).reshape(-1, 3)
    assert len(lp) == len(lt) and len(lp) % 2 == 0
    rec = np.ascontiguousarray(cam_rec, dtype=np.float32)
    out = np.empty((H, W, 3), np.float32)
    prim = np.empty((H, W), np.int32)
    L = lib()
    L.ora_render_strands.restype = ctypes.c_long
    L.ora_render_strands.argtypes = [c_f, c_f, ctypes.c_int, c_i, ctypes.c_int, c_f, c_f, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_float, c_f, c_i]
    n = L.ora_render_strands(_p(rec), _p(verts), len(verts), _p(faces, c_i), len(faces), _p(lp), _p(lt), len(lp) // 2, H,
                             W, pixel_center, width, int(line_rule), color_option, depth_option, clear, _p(out),
                             _p(prim, c_i))
    if n < 0:
        raise MemoryError("ora_render_strands")
    return out, prim, int(n)
