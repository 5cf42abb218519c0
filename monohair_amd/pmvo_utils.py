"""Host-side helpers of the PMVO path -- the mirror of the parts of the reference's `Utils/PMVO_utils.py`
that PMVO.py uses: loaders (:255-362), compute_points_similarity (:366-382), p2v (:386-404), voxel<->world
(:407-421), .mat readers (:86-113) and .hair IO (:47-83, :662-680).

Arithmetic that matters runs in HIP kernels (medoid consensus) or is a two-line numpy formula restated from
the reference; image decoding uses PIL (OpenCV is not a dependency here), meshes are read with a small OBJ
reader.  Names and argument order follow the reference so that PMVO.py reads the same.
"""
import math
import os
import sys
import struct

import numpy as np
import torch

from . import _lib

VOXEL_MIN = np.array([-0.32, -0.32, -0.24])
VOXEL_SIZE = 0.005 / 2
GRID_RESOLUTION = np.array([256, 256, 192]).astype(np.int32)


# ----------------------------------------------------------------------------------------- images / maps
def _imread_gray(path):
    from PIL import Image

    return np.array(Image.open(path).convert("L"))


def _imread_bgr(path):
    from PIL import Image

    return np.array(Image.open(path).convert("RGB"))[..., ::-1].copy()


def _find(path_dir, view, suffixes=(".png", ".JPG", ".jpg", ".jpeg", ".PNG")):
    for s in suffixes:
        p = os.path.join(path_dir, view + s)
        if os.path.exists(p):
            return p
    raise FileNotFoundError("no image for view %r under %s" % (view, path_dir))


def Load_Ori_And_Conf(camera, Ori_path, Conf_path):
    """best_ori/<view>: gray u8 = orientation in degrees -> theta' = (180-pix)/180*pi -> (sin, cos) float64;
    conf/<view>: gray u8 / 255 (PMVO_utils.py:255-276).  The reference's suffix sniffing (:258-264) only works
    on case-insensitive file systems for .JPG data; we look for .png, .JPG, .jpg in that order."""
    Ori, Conf = {}, {}
    for view, _ in camera.items():
        # `180 - o` is evaluated on the uint8 image cv2.imread returns (PMVO_utils.py:265-266): pixel codes above 180
        # wrap modulo 256 (code 200 -> 236 degrees), they do not go negative
        o = (np.uint8(180) - _imread_gray(_find(Ori_path, view)).astype(np.uint8)).astype(np.uint8)
        o = o / 180 * math.pi
        Ori[view] = np.stack([np.sin(o), np.cos(o)], -1)
        Conf[view] = _imread_gray(_find(Conf_path, view)) / 255.0
    return Ori, Conf


def load_depth(camera, path, type="npy"):
    """render_depth/<view>.npy, float32 [H,W,3] (value = -z_cam/2*255, background 255) (PMVO_utils.py:278-295)."""
    return {view: np.load(os.path.join(path, view + ".npy")).astype(np.float32) for view, _ in camera.items()}


def load_mask(camera, path):
    """hair_mask/<view>: BGR u8, values < 50 zeroed, /255 (PMVO_utils.py:297-313)."""
    masks = {}
    for view, _ in camera.items():
        mask = _imread_bgr(_find(path, view))
        mask[mask < 50] = 0
        masks[view] = mask / 255.0
    return masks


def map_code_lut():
    """[256,4] float32 table {ori_row, ori_col, conf, mask}: what Load_Ori_And_Conf / load_mask above give for each
    8-bit pixel code (the same float64 numpy expressions, then the float32 cast of PMVO.__init__, PMVO.py:23-26)."""
    code = np.arange(256, dtype=np.uint8)
    o = (np.uint8(180) - code).astype(np.uint8) / 180 * math.pi      # uint8 arithmetic: codes > 180 wrap (see above)
    m = code.copy()
    m[m < 50] = 0
    return np.stack([np.sin(o), np.cos(o), code / 255.0, m / 255.0], -1).astype(np.float32)


def _threaded(fn, items, threads):
    if threads <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(threads) as ex:      # PIL and numpy IO release the GIL while decoding
        return list(ex.map(fn, items))


def load_maps_u8(camera, Ori_path, Conf_path, mask_path, threads=8):
    """The three 8-bit map sets as pixel codes (no float64 decode on the host): dicts view -> uint8 [H,W] of
    best_ori/ (gray), conf/ (gray) and hair_mask/ (channel 0 of the BGR image, which is all PMVO reads,
    PMVO.py:523).  PMVO.from_u8 decodes them on the GPU through map_code_lut()."""
    views = list(camera.keys())

    def one(view):
        return (_imread_gray(_find(Ori_path, view)), _imread_gray(_find(Conf_path, view)),
                np.ascontiguousarray(_imread_bgr(_find(mask_path, view))[..., 0]))

    got = _threaded(one, views, threads)
    return ({v: g[0] for v, g in zip(views, got)}, {v: g[1] for v, g in zip(views, got)},
            {v: g[2] for v, g in zip(views, got)})


def load_depth_plane(camera, path, threads=8):
    """render_depth/<view>.npy -> channel 0 only, float32 [H,W] (the only channel PMVO reads, PMVO.py:485)."""
    views = list(camera.keys())

    def one(view):
        d = np.load(os.path.join(path, view + ".npy"), mmap_mode="r")
        return np.ascontiguousarray(d[..., 0] if d.ndim == 3 else d, dtype=np.float32)

    return dict(zip(views, _threaded(one, views, threads)))


# ----------------------------------------------------------------------------------------- meshes / points
def _read_obj_slow(lines):
    vs, fs = [], []
    for line in lines:
        if line.startswith(b"v "):
            p = line.split()
            vs.append([float(p[1]), float(p[2]), float(p[3])])
        elif line.startswith(b"f "):
            idx = [int(t.split(b"/")[0]) for t in line.split()[1:]]
            idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
            for k in range(1, len(idx) - 1):
                fs.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(vs, dtype=np.float64).reshape(-1, 3), np.asarray(fs, dtype=np.int64).reshape(-1, 3)


def read_obj(path):
    """Minimal Wavefront OBJ reader: (vertices [N,3] f64, faces [M,3] int) -- triangles / fan-triangulated polys.
    Meshes of the pipeline have 10^5..10^6 lines, so the regular case (every `v` line with the same number of
    fields, every `f` line a triangle with positive indices) is parsed in bulk; anything else takes the line loop."""
    import re

    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    vl = [ln for ln in lines if ln.startswith(b"v ")]
    fl = [ln for ln in lines if ln.startswith(b"f ")]
    try:
        nv = len(vl)
        vt = np.fromstring(b" ".join(ln[2:] for ln in vl), dtype=np.float64, sep=" ")      # C text parser
        if nv == 0 or vt.size % nv != 0 or vt.size // nv < 3:
            raise ValueError
        verts = np.ascontiguousarray(vt.reshape(nv, -1)[:, :3])
        if fl:
            ft = np.fromstring(re.sub(rb"/[^ \t\r]*", b"", b" ".join(ln[2:] for ln in fl)), dtype=np.int64, sep=" ")
            if ft.size != 3 * len(fl):
                raise ValueError                    # polygons: fan triangulation in the line loop
            faces = ft.reshape(len(fl), 3)
            if faces.min() < 1:
                raise ValueError                    # relative (negative) indices
            faces = faces - 1
        else:
            faces = np.zeros((0, 3), np.int64)
        return verts, faces
    except (ValueError, DeprecationWarning):
        return _read_obj_slow(lines)


def vertex_normals(vertices, faces):
    n = np.zeros_like(vertices)
    if len(faces):
        fn = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]], vertices[faces[:, 2]] - vertices[faces[:, 0]])
        for k in range(3):
            np.add.at(n, faces[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.maximum(ln, 1e-12)


def load_bust(path):
    """(vertices, faces, normals) of the bust mesh (PMVO_utils.py:176-181)."""
    v, f = read_obj(path)
    return v, f, vertex_normals(v, f)


def sample_points_uniformly(vertices, faces, number_of_points, rng=None):
    """Area-weighted uniform surface sampling (what open3d's sample_points_uniformly does at PMVO_utils.py:346;
    its RNG is not controlled by the reference's seed either, so this step is unpinned by construction)."""
    rng = np.random.default_rng(0) if rng is None else rng
    if len(faces) == 0:
        return vertices[rng.integers(0, len(vertices), number_of_points)]
    a, b, c = vertices[faces[:, 0]], vertices[faces[:, 1]], vertices[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    tri = rng.choice(len(faces), size=number_of_points, p=area / area.sum())
    r1 = np.sqrt(rng.random(number_of_points))[:, None]
    r2 = rng.random(number_of_points)[:, None]
    return (1 - r1) * a[tri] + r1 * (1 - r2) * b[tri] + r1 * r2 * c[tri]


def SamplePointsAroundmesh(colmap_points, bbox_min, vsize, num_per_grid=32, grid_resolution=[512, 512, 384]):
    """PMVO_utils.py:316-339: occupied voxels of the point set (y,z flipped), num_per_grid uniform jitters each
    (np.random, seeded by options.process_options like the reference), flipped back."""
    g = np.asarray(grid_resolution)
    colmap_points = colmap_points.copy()
    colmap_points[:, 1:] *= -1
    idx = np.round((colmap_points - bbox_min) / vsize).astype(np.int32)
    idx = np.clip(idx, 0, g - 1)
    # the occupied cells in np.nonzero order of the reference's dense grid (x, then y, then z ascending), without
    # the 100 M-cell grid itself: the sorted unique linear indices
    lin = np.unique((idx[:, 0].astype(np.int64) * g[1] + idx[:, 1]) * g[2] + idx[:, 2])
    indices = np.stack([lin // (g[1] * g[2]), (lin // g[2]) % g[1], lin % g[2]], 1)
    base = np.concatenate([indices] * num_per_grid, 0)
    sample = (base + np.random.random(base.shape)) * vsize + bbox_min
    sample[:, 1:] *= -1
    return sample


def load_colmap_points(path, bbox_min, bust_to_origin, vsize=0.005, grid_resolution=[128, 128, 96], sample=True,
                       num_per_grid=8):
    """PMVO_utils.py:341-362."""
    v, f = read_obj(path)
    print("num_p:", v.shape[0])
    pts = sample_points_uniformly(v, f, v.shape[0] * 5)
    pts = pts + bust_to_origin
    if sample:
        out = SamplePointsAroundmesh(pts.copy(), np.asarray(bbox_min), vsize, num_per_grid=num_per_grid,
                                     grid_resolution=grid_resolution)
        print("num sample:", out.shape[:])
        return out
    return pts


# ----------------------------------------------------------------------------------------- consensus (HIP)
def _ctx_for(device):
    """A bare library context for kernels that need no views (medoid, Gabor)."""
    import ctypes

    key = str(device)
    if key not in _ctx_for.cache:
        h = ctypes.c_void_p()
        idx = torch.device(device).index or 0
        _lib.check(_lib.lib().mh_ctx_create(idx, ctypes.byref(h)), "mh_ctx_create")
        _ctx_for.cache[key] = h
    return _ctx_for.cache[key]


_ctx_for.cache = {}


def compute_points_similarity(ori, return_index=False):
    """PMVO_utils.py:366-382: ori [N,K,3] device tensor -> medoid orientation [N,3] (HIP kernel mh_medoid_dense)."""
    if not ori.is_cuda:
        raise _lib.MhError("compute_points_similarity runs on the GPU only (no CPU fallback)")
    ori = ori.type(torch.float).contiguous()
    N, K, _ = ori.shape
    out = torch.empty((N, 3), dtype=torch.float32, device=ori.device)
    idx = torch.empty((N,), dtype=torch.int32, device=ori.device)
    with torch.cuda.device(ori.device):
        _lib.check(_lib.lib().mh_medoid_dense(_ctx_for(ori.device), _lib.ptr(ori), N, K, _lib.ptr(out), _lib.ptr(idx),
                                              _lib.stream_ptr()), "mh_medoid_dense")
    return (out, idx) if return_index else out


class GridKNN:
    """Exact k-nearest-neighbour search on the GPU (csrc/knn.hip) with scipy.spatial.KDTree's query semantics:
    neighbours sorted by distance (fp64 from the fp32 coordinates), self included when the query is a data point.
    Used for the two `points_tree.query(sub_points, 100)` of refine (PMVO.py:612,671).  The uniform grid only
    affects speed: queries the kernel cannot finish with one cell size (candidate buffer overflow in dense spots,
    ring limit in sparse ones) are re-run on a finer / coarser grid, still on the GPU."""

    def __init__(self, points, k_hint=100, device="cuda:0"):
        self.device = torch.device(device)
        if isinstance(points, torch.Tensor):      # float32 [M,3] already on the device (the drivers' own copy)
            self._raw = points.to(self.device).type(torch.float).contiguous().reshape(-1, 3)
        else:
            pts_np = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
            self._raw = torch.from_numpy(pts_np).to(self.device)
        self.M = int(self._raw.shape[0])
        # bounding box: numpy's axis-0 reduction of an [M,3] array costs 2-5 ms per call at 3e5 points (inner loop of 3),
        # the device needs two tiny launches
        if self.M:
            box = torch.empty(6, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().mh_points_bbox(_ctx_for(self.device), _lib.ptr(self._raw), self.M, _lib.ptr(box),
                                                     _lib.stream_ptr()), "mh_points_bbox")
            box = box.cpu().numpy()
            self._lo, self._hi = box[:3].copy(), box[3:].copy()
        else:
            self._lo, self._hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        ext = float((self._hi - self._lo).max()) + 1e-6
        self._ext = ext
        self._grids = {}
        self._scratch = torch.empty(int(_lib.lib().mh_grid_scratch_bytes(self.M)), dtype=torch.uint8,
                                    device=self.device)
        self._nocc = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._order_tmp = torch.empty(self.M, dtype=torch.int32, device=self.device)

        def occupancy(h):       # mean number of points per non-empty cell at cell size h
            grid, dims = self._geometry(h)
            self._build(grid, dims, None, self._order_tmp, None, self._nocc)
            return self.M / float(max(int(self._nocc.item()), 1))

        # local dimension and density from the mean occupancy at two scales -> radius of the k-ball -> cell size
        h0 = max(ext / 128.0, 1e-6)
        c1, c2 = occupancy(h0), occupancy(2 * h0)
        dim = min(3.0, max(1.0, math.log(max(c2 / c1, 1.01), 2)))
        ball = {1: 2.0, 2: math.pi, 3: 4.18879}[int(round(dim))]
        rk = h0 * (max(k_hint, 1) / (c1 * ball)) ** (1.0 / dim)
        # cells a little larger than the radius of the k-ball: the 27 cells around a query hold every point within one
        # cell size of it, so the first ring answers almost every query and only the points within that reach get sorted
        self.h = max(rk * 1.2, ext / 480.0)
        self.last_retries = 0

    def _geometry(self, h):
        origin = self._lo.astype(np.float32)
        dims = np.floor((self._hi.astype(np.float64) - origin) / h).astype(np.int64) + 1
        return (np.array([origin[0], origin[1], origin[2], h], dtype=np.float32),
                np.maximum(dims, 1).astype(np.int32))

    def _build(self, grid, dims, pts_sorted, order, start, nocc):
        import ctypes

        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_grid_build(
                _ctx_for(self.device), grid.ctypes.data_as(ctypes.c_void_p), dims.ctypes.data_as(ctypes.c_void_p),
                _lib.ptr(self._raw), self.M, _lib.ptr(self._scratch), self._scratch.numel(), _lib.ptr(pts_sorted),
                _lib.ptr(order), _lib.ptr(start), _lib.ptr(nocc), _lib.stream_ptr()), "mh_grid_build")

    def _grid(self, h):
        """(origin+h [4] f32 host, dims [3] i32 host, points sorted by cell, original indices, cell_start) -- built by
        mh_grid_build (csrc/sortgroup.hip): cell keys, stable radix sort, gather, first position of every cell."""
        h = float(np.float32(max(h, self._ext / 480.0)))
        if h not in self._grids:
            grid, dims = self._geometry(h)
            ncell = int(dims[0]) * int(dims[1]) * int(dims[2])
            pts = torch.empty((self.M, 3), dtype=torch.float32, device=self.device)
            order = torch.empty(self.M, dtype=torch.int32, device=self.device)
            start = torch.empty(ncell + 1, dtype=torch.int32, device=self.device)
            self._build(grid, dims, pts, order, start, None)
            self._grids[h] = (grid, dims, pts, order, start)
        return self._grids[h]

    def _run(self, h, q, k, perm=None, valid=None, zero=False):
        import ctypes

        grid, dims, pts, order, start = self._grid(h)
        Q = q.shape[0]
        # (zero: rows of queries the kernel does not finish stay valid indices -- a caller that reads the table before it has
        # looked at the status must not gather through uninitialised memory)
        out = (torch.zeros if zero else torch.empty)((Q, k), dtype=torch.int32, device=self.device)
        status = torch.empty((Q,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_knn_grid(_ctx_for(self.device), grid.ctypes.data_as(ctypes.c_void_p),
                                              dims.ctypes.data_as(ctypes.c_void_p), _lib.ptr(pts), _lib.ptr(order),
                                              _lib.ptr(start), _lib.ptr(q), 1 if q.dtype == torch.float64 else 0, Q, k,
                                              1, _lib.ptr(perm), _lib.ptr(valid), _lib.ptr(out), _lib.ptr(status),
                                              _lib.stream_ptr()),
                       "mh_knn_grid")
        return out, status

    def cell_order(self):
        """int32 [M] on the device: the data points' indices sorted by cell of the search grid (spatial neighbours adjacent)."""
        return self._grid(self.h)[3]

    def record_stream(self, stream):
        """The grid was built on another stream than the one that will query it from now on (refine adopts the grid
        optimize's prefetch built): tell the caching allocator."""
        for t in (self._raw, self._scratch, self._nocc, self._order_tmp):
            t.record_stream(stream)
        for g in self._grids.values():
            for t in g[2:]:
                t.record_stream(stream)

    def _queries(self, queries):
        """[Q,3] queries on the device in the precision they were given in (float64 stays float64, see query)."""
        if isinstance(queries, torch.Tensor):
            q = queries.to(self.device)
            return (q if q.dtype == torch.float64 else q.type(torch.float)).contiguous().reshape(-1, 3)
        qn = np.ascontiguousarray(queries)
        if qn.dtype != np.float64:
            qn = qn.astype(np.float32, copy=False)
        return torch.from_numpy(qn.reshape(-1, 3)).to(self.device).contiguous()

    def query_nosync(self, queries, k, valid_dev=None, self_query=False):
        """The first attempt of query() only, nothing read back: -> (index [Q,k] int32, status [Q] int32) on the device.
        status != 0 marks queries that need query()'s retries on another cell size (their rows of the index are zeros).
        With valid_dev (uint8 [M] on the device) the caller guarantees that at least k points are flagged (fewer: the kernel
        pads with -1, which is not an index).  refine's device-resident pass checks the status once it has synchronised
        anyway and falls back to query() if any is set."""
        q = self._queries(queries)
        perm = self._grid(self.h)[3] if (self_query and q.shape[0] == self.M) else None
        return self._run(self.h, q, min(int(k), self.M), perm, valid_dev, zero=True)

    def _finish(self, q, k, out, st, valid):
        """The retries of a query: rows of `out` whose status `st` (host array, updated in place) is not 0 are searched again
        on finer (candidate buffer overflow) or coarser (ring limit) cells, what is left after that exhaustively -- all on
        the GPU, the loop on the host."""
        self.last_retries = 0
        for code, factor in ((2, 0.5), (1, 2.0)):        # overflow -> finer cells, ring limit -> coarser cells
            h = self.h
            for _ in range(6):
                bad = np.flatnonzero(st == code)
                if not bad.size:
                    break
                h *= factor
                bad_d = torch.from_numpy(bad).to(self.device)
                o2, s2 = self._run(h, q[bad_d].contiguous(), k, None, valid)
                out[bad_d] = o2
                st[bad] = s2.cpu().numpy()
                self.last_retries += 1
        # extreme density contrast (a sparse query whose k-ball swallows a dense cluster) defeats every cell size:
        # those few queries are answered by exhaustive search, also on the GPU (same fp64 distances, stable sort =
        # ties by index)
        left = np.flatnonzero(st != 0)
        self.last_exhaustive = int(left.size)
        if left.size:
            p64 = self._raw.to(torch.float64)
            for i in left.tolist():
                d = p64 - q[i].to(torch.float64)
                d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
                if valid is not None:
                    d2 = torch.where(valid.bool(), d2, torch.full_like(d2, float("inf")))
                out[i] = torch.sort(d2, stable=True).indices[:k].to(torch.int32)

    def finish_nosync(self, queries, k, out, status_host, valid_dev=None):
        """query()'s retries for the rows a query_nosync left unfinished (status_host != 0), in place in `out`."""
        self._finish(self._queries(queries), min(int(k), self.M), out, np.array(status_host, copy=True), valid_dev)

    def query(self, queries, k, int32=False, self_query=False, valid=None):
        """-> index [Q,k] device tensor, int64 (or int32 as the kernel writes it); k clamped to the number of points,
        like the drivers do.  Queries keep their precision: float64 arrays (the shell points of refine, PMVO.py:671) are
        searched with their exact coordinates, as scipy does.  self_query=True (queries are the data points, in the same
        order): the waves take the queries in cell order, so neighbouring waves read the same cells.
        valid (bool [M]): only these data points count as neighbours -- the answer equals
        KDTree(points[valid]).query(...) with the indices mapped back to `points` (order-preserving, so the
        (distance, index) order is the same); k is clamped to the number of valid points."""
        if valid is not None:
            vh = np.ascontiguousarray(valid, dtype=np.uint8)
            k = min(int(k), int(vh.sum()))
            valid = torch.from_numpy(vh).to(self.device)
        k = min(int(k), self.M)
        q = self._queries(queries)
        perm = self._grid(self.h)[3] if (self_query and q.shape[0] == self.M) else None
        out, status = self._run(self.h, q, k, perm, valid)
        self._finish(q, k, out, status.cpu().numpy(), valid)        # (host side: no torch kernels on this path)
        return out if int32 else out.long()


def spatial_order(points_dev, cell=0.005, half_extent=2.56):
    """Rows of float32 [M,3] device points sorted by the cell of a FIXED uniform grid (cells of `cell` metres over
    +-`half_extent`: 1024^3 cells at the defaults; points outside are clamped into the border cells) -> int32 [M] permutation on
    the device (mh_grid_build: cell keys + one radix sort; nothing is read back).  Only the ORDER in which a kernel takes the
    rows depends on it (mh_filter_points_ordered): 64 consecutive entries are spatial neighbours."""
    import ctypes

    dev = points_dev.device
    M = int(points_dev.shape[0])
    order = torch.empty(M, dtype=torch.int32, device=dev)
    if M == 0:
        return order
    L = _lib.lib()
    n = int(round(2 * half_extent / cell))
    grid = np.array([-half_extent, -half_extent, -half_extent, cell], dtype=np.float32)
    dims = np.array([n, n, n], dtype=np.int32)
    scratch = torch.empty(int(L.mh_grid_scratch_bytes(M)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.mh_grid_build(_ctx_for(dev), grid.ctypes.data_as(ctypes.c_void_p), dims.ctypes.data_as(ctypes.c_void_p),
                                   _lib.ptr(points_dev), M, _lib.ptr(scratch), scratch.numel(), None, _lib.ptr(order), None,
                                   None, _lib.stream_ptr()), "mh_grid_build")
    return order


def p2v(points, voxel_min, voxel_size, grid_resolution):
    """PMVO_utils.py:386-404: flips y,z IN PLACE (mutates the caller's array, like the reference), float64
    round-half-even, clip.  Returns (x, y, z) int32 arrays."""
    points[:, 1:] *= -1
    idx = np.round((points - voxel_min) / voxel_size).astype(np.int32)
    g = np.asarray(grid_resolution)
    x = np.clip(idx[:, 0], 0, g[0] - 1)
    y = np.clip(idx[:, 1], 0, g[1] - 1)
    z = np.clip(idx[:, 2], 0, g[2] - 1)
    return x, y, z


def voxel_fit(select_points, select_ori, device, voxel_min=VOXEL_MIN, voxel_size=VOXEL_SIZE,
              grid_resolution=GRID_RESOLUTION, dense=True):
    """The volume fit of refine (PMVO.py:695-726) on the GPU: one call (mh_voxel_group) evaluates p2v in float64, sorts the
    voxel keys stably (point order inside a voxel is preserved, which is what the reference's dict of lists does) and
    gathers the sign-canonicalised orientations (ori.y > 0 -> negated) in that order; one segmented-medoid launch for all
    voxels.  Only the run boundaries of the sorted keys are found on the host.

    Returns dict(voxels [G,3] int64 (x,y,z), ori [G,3] f32) and, with dense=True, occ [X,Y,Z] / ori [X,Y,Z,3]
    float64 numpy arrays as the reference builds them.  The inputs are NOT modified (the reference flips them in place,
    PMVO.py:697-698 and p2v; nothing reads them afterwards)."""
    import ctypes

    g = np.asarray(grid_resolution).astype(np.int64)
    dev = torch.device(device)
    pts = np.ascontiguousarray(select_points).reshape(-1, 3)
    if pts.dtype != np.float64:
        pts = pts.astype(np.float32, copy=False)
    n = int(pts.shape[0])
    L = _lib.lib()
    if n:
        pd = torch.from_numpy(pts).to(dev)
        od = torch.from_numpy(np.ascontiguousarray(select_ori, dtype=np.float32).reshape(-1, 3)).to(dev)
        ks_d = torch.empty(n, dtype=torch.int64, device=dev)      # non-negative: the same bits as the u64 keys
        order_d = torch.empty(n, dtype=torch.int32, device=dev)
        o = torch.empty((n, 3), dtype=torch.float32, device=dev)
        scratch = torch.empty(int(L.mh_voxel_group_scratch_bytes(n)), dtype=torch.uint8, device=dev)
        vmin = np.ascontiguousarray(voxel_min, dtype=np.float64)
        dims = np.ascontiguousarray(g, dtype=np.int32)
        with torch.cuda.device(dev):
            _lib.check(L.mh_voxel_group(_ctx_for(dev), _lib.ptr(pd), 1 if pts.dtype == np.float64 else 0, _lib.ptr(od), n,
                                        vmin.ctypes.data_as(ctypes.c_void_p), float(voxel_size),
                                        dims.ctypes.data_as(ctypes.c_void_p), _lib.ptr(scratch), scratch.numel(),
                                        _lib.ptr(ks_d), _lib.ptr(order_d), _lib.ptr(o), _lib.stream_ptr()),
                       "mh_voxel_group")
        ks = ks_d.cpu().numpy()
    else:
        ks = np.zeros(0, np.int64)
        o = torch.empty((0, 3), dtype=torch.float32, device=dev)
    starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]])) if n else np.zeros(0, np.int64)
    G = int(starts.size)
    seg_h = np.concatenate([starts, [n]]).astype(np.int32)
    max_group = int(np.diff(seg_h).max()) if G else 0
    seg = torch.from_numpy(seg_h).to(dev)
    med = torch.empty((G, 3), dtype=torch.float32, device=dev)
    if G:
        with torch.cuda.device(dev):
            _lib.check(L.mh_medoid_segmented(_ctx_for(dev), _lib.ptr(o), _lib.ptr(seg), G, max_group,
                                             _lib.ptr(med), None, _lib.stream_ptr()), "mh_medoid_segmented")
    kv = ks[starts] if G else np.zeros(0, np.int64)
    vox = torch.from_numpy(np.stack([kv // (int(g[1]) * int(g[2])), (kv // int(g[2])) % int(g[1]), kv % int(g[2])],
                                    1).astype(np.int64).reshape(-1, 3))
    out = dict(voxels=vox, ori=med)
    if dense:
        occ = np.zeros(tuple(g))
        ori = np.zeros(tuple(g) + (3,))
        v = vox.cpu().numpy()
        occ[v[:, 0], v[:, 1], v[:, 2]] = 1
        ori[v[:, 0], v[:, 1], v[:, 2]] = med.cpu().numpy().astype(np.float64)
        out["occ"], out["ori_dense"] = occ, ori
    return out


def voxel_fit_device(pts_dev, ori_dev, n, device, voxel_min=VOXEL_MIN, voxel_size=VOXEL_SIZE, grid_resolution=GRID_RESOLUTION):
    """voxel_fit for rows [0, n) of float32 [*,3] tensors that are ALREADY on the device (refine's device-resident pass):
    keys + stable sort + canonicalised gather (mh_voxel_group), run boundaries on the device (mh_segment_heads), one
    segmented-medoid launch; the host reads two counters in between and the G voxel keys + orientations at the end.
    -> (voxels [G,3] int64 (x,y,z) ascending, ori [G,3] float32) numpy arrays -- the same values as voxel_fit."""
    import ctypes

    g = np.asarray(grid_resolution).astype(np.int64)
    dev = torch.device(device)
    n = int(n)
    if n == 0:
        return np.zeros((0, 3), np.int64), np.zeros((0, 3), np.float32)
    L = _lib.lib()
    ctx = _ctx_for(dev)
    ks_d = torch.empty(n, dtype=torch.int64, device=dev)
    order_d = torch.empty(n, dtype=torch.int32, device=dev)
    o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    scratch = torch.empty(int(L.mh_voxel_group_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    seg = torch.empty(n + 1, dtype=torch.int32, device=dev)
    heads = torch.empty(n, dtype=torch.int64, device=dev)
    meta = torch.empty(2, dtype=torch.int32, device=dev)
    sel_scratch = torch.empty(int(L.mh_select_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    vmin = np.ascontiguousarray(voxel_min, dtype=np.float64)
    dims = np.ascontiguousarray(g, dtype=np.int32)
    with torch.cuda.device(dev):
        st = _lib.stream_ptr()
        _lib.check(L.mh_voxel_group(ctx, _lib.ptr(pts_dev), 0, _lib.ptr(ori_dev), n, vmin.ctypes.data_as(ctypes.c_void_p),
                                    float(voxel_size), dims.ctypes.data_as(ctypes.c_void_p), _lib.ptr(scratch),
                                    scratch.numel(), _lib.ptr(ks_d), _lib.ptr(order_d), _lib.ptr(o), st), "mh_voxel_group")
        _lib.check(L.mh_segment_heads(ctx, _lib.ptr(ks_d), n, _lib.ptr(seg), _lib.ptr(heads), _lib.ptr(meta),
                                      _lib.ptr(sel_scratch), sel_scratch.numel(), st), "mh_segment_heads")
        G, max_group = (int(v) for v in meta.cpu().numpy())          # (the one synchronisation of the fit)
        med = torch.empty((G, 3), dtype=torch.float32, device=dev)
        _lib.check(L.mh_medoid_segmented(ctx, _lib.ptr(o), _lib.ptr(seg), G, max(max_group, 1), _lib.ptr(med), None, st),
                   "mh_medoid_segmented")
        kv = heads[:G].cpu().numpy()
        vori = med.cpu().numpy()
    vox = np.stack([kv // (int(g[1]) * int(g[2])), (kv // int(g[2])) % int(g[1]), kv % int(g[2])], 1).astype(np.int64)
    return vox.reshape(-1, 3), vori


class NpyRowWriter:
    """The bytes of `np.save(path, a)` for a C-contiguous array of known shape, written a block of leading rows at a time
    (version 1.0 header, then the raw rows): optimize() streams its result files out while later chunks still compute."""

    def __init__(self, path, shape, dtype):
        self._f = open(path, "wb")
        np.lib.format.write_array_header_1_0(self._f, {"descr": np.lib.format.dtype_to_descr(np.dtype(dtype)),
                                                       "fortran_order": False, "shape": tuple(int(v) for v in shape)})

    def write(self, rows):
        rows = np.ascontiguousarray(rows)
        self._f.write(rows.view(np.uint8).reshape(-1).data if rows.size else b"")

    def close(self):
        self._f.close()


def save_ori_occ_mat(path, occ, ori):
    """Layout + scipy.io.savemat of PMVO.py:753-764: Ori [Y,X,3*Z] (last index c*Z+z), Occ [Y,X,Z], float64."""
    g = occ.shape
    o = ori.transpose((0, 1, 3, 2)).reshape(g[0], g[1], g[2] * 3).transpose((1, 0, 2))
    import scipy.io          # (lazy: 0.15 s of import time the command line does not need)

    scipy.io.savemat(os.path.join(path, "Ori3D.mat"), {"Ori": o})
    scipy.io.savemat(os.path.join(path, "Occ3D.mat"), {"Occ": occ.transpose((1, 0, 2))})


def _mat5_prefix(name, dims):
    """Bytes of a MAT-v5 file up to the float64 payload of ONE real double array `name` of shape `dims` (the layout
    scipy.io.savemat writes: 128-byte header, miMATRIX{array flags, dimensions, name, real part})."""
    import struct
    import time

    nb = name.encode()
    assert 1 <= len(nb) <= 4, "small-element name form"
    ndata = 8 * int(np.prod(dims))
    assert ndata < 2 ** 32 - 64, "MAT v5 elements are limited to 4 GiB"
    text = ("MATLAB 5.0 MAT-file Platform: posix, Created on: %s" % time.asctime()).encode()
    head = text.ljust(124, b"\x00") + struct.pack("<H", 0x0100) + b"IM"
    flags = struct.pack("<IIII", 6, 8, 6, 0)                                   # miUINT32, mxDOUBLE_CLASS
    d = struct.pack("<II", 5, 4 * len(dims)) + struct.pack("<%di" % len(dims), *[int(x) for x in dims])
    d += b"\x00" * (-len(d) % 8)
    nm = struct.pack("<HH", 1, len(nb)) + nb.ljust(4, b"\x00")                  # miINT8, small data element
    real = struct.pack("<II", 9, ndata)                                        # miDOUBLE
    body = flags + d + nm + real
    return head + struct.pack("<II", 14, len(body) + ndata + (-ndata % 8)) + body, ndata


def save_ori_occ_mat_sparse(path, grid_resolution, voxels, ori, threads=1):
    """Same two files as save_ori_occ_mat (PMVO.py:753-764) written from the occupied voxels only: `voxels` [G,3]
    (x,y,z) and `ori` [G,3]; later rows win on duplicates, as the reference's fancy assignments do (:746-747).
    The float64 payload (Ori [Y,X,3Z] with last index c*Z+z, Occ [Y,X,Z], column-major as MAT v5 stores it) is
    created as a zero-filled sparse file and only the occupied elements are stored (mh_mat_write_sparse: a shared
    mapping; the two files are written concurrently, one thread each -- measured on the GPU box, more threads per file
    only contend for the page cache: 19 ms with one, 34 ms with sixteen, tools/bench_mat.py)."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor

    X, Y, Z = (int(v) for v in grid_resolution)
    v = np.asarray(voxels, dtype=np.int64).reshape(-1, 3)
    o = np.ascontiguousarray(np.asarray(ori, dtype=np.float64).reshape(-1, 3))
    lin = v[:, 1] + Y * (v[:, 0] + X * v[:, 2])                                # Occ[y,x,z], y fastest
    L = _lib.lib()

    def write(fname, name, dims, idx, val):
        prefix, ndata = _mat5_prefix(name, dims)
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        val = np.ascontiguousarray(val, dtype=np.float64)
        _lib.check(L.mh_mat_write_sparse(os.path.join(path, fname).encode(), prefix, len(prefix), ndata,
                                         idx.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p),
                                         len(idx), threads), "mh_mat_write_sparse")

    jobs = (("Occ3D.mat", "Occ", (Y, X, Z), lin, np.ones(len(v))),
            ("Ori3D.mat", "Ori", (Y, X, 3 * Z), (lin[:, None] + (Y * X * Z) * np.arange(3)[None, :]).reshape(-1),
             o.reshape(-1)))
    with ThreadPoolExecutor(2) as pool:           # ctypes releases the GIL during the call
        list(pool.map(lambda j: write(*j), jobs))


class SparseMatWriter:
    """Ori3D.mat / Occ3D.mat of save_ori_occ_mat_sparse written in two phases: the constructor creates and maps both files
    and -- on a background thread per file -- makes the pages resident that the voxels of `candidate_points` fall on (every
    point that can end up in the volume: the page faults of the zero-filled mappings, 16-19 ms per pass, are taken while the
    GPU works on the rest of refine); finish(voxels, ori) stores the occupied elements and closes.  Same bytes as
    save_ori_occ_mat_sparse."""

    def __init__(self, path, grid_resolution, candidate_points=None, voxel_min=VOXEL_MIN, voxel_size=VOXEL_SIZE):
        import ctypes
        import threading

        self.X, self.Y, self.Z = (int(v) for v in grid_resolution)
        X, Y, Z = self.X, self.Y, self.Z
        self._L = _lib.lib()
        self._files = [None, None]
        self._names = [None, None]
        self._error = []          # errors of the pre-fault hint (they cost time, not correctness)
        self._fatal = []          # errors of creating the files: re-raised by finish()

        def one_file(k, fname, name, dims, lin):
            try:
                prefix, ndata = _mat5_prefix(name, dims)
                h = ctypes.c_void_p()
                # (creating the file truncates an existing one: releasing a previous run's 100-300 MB of page cache takes
                # tens of milliseconds -- also on this thread)
                # under a temporary name, renamed when the elements are in: a pass that fails half-way leaves the files of
                # an earlier run alone
                _lib.check(self._L.mh_mat_sparse_open((os.path.join(path, fname) + self._TMP).encode(), prefix,
                                                      len(prefix), ndata, ctypes.byref(h)), "mh_mat_sparse_open")
                self._files[k] = h
                self._names[k] = os.path.join(path, fname)
            except BaseException as e:
                self._fatal.append(e)
                return
            try:
                if lin is not None and len(lin):
                    idx = lin if k == 0 else (lin[None, :] + (Y * X * Z) * np.arange(3, dtype=np.int64)[:, None]).reshape(-1)
                    idx = np.ascontiguousarray(idx, dtype=np.int64)
                    self._L.mh_mat_sparse_touch(h, idx.ctypes.data_as(ctypes.c_void_p), len(idx))
            except BaseException as e:
                self._error.append(e)

        def both():
            lin = None
            try:
                if candidate_points is not None and len(candidate_points):
                    pts = np.array(candidate_points, dtype=np.float64, copy=True)       # (p2v flips its argument in place)
                    x, y, z = p2v(pts, np.asarray(voxel_min, dtype=np.float64), voxel_size, (X, Y, Z))
                    lin = y.astype(np.int64) + Y * (x.astype(np.int64) + X * z.astype(np.int64))   # (touch skips repeated pages)
            except BaseException as e:
                self._error.append(e)
            t2 = threading.Thread(target=one_file, args=(1, "Ori3D.mat", "Ori", (Y, X, 3 * Z), lin))
            t2.start()
            one_file(0, "Occ3D.mat", "Occ", (Y, X, Z), lin)
            t2.join()

        t = threading.Thread(target=both, daemon=True)
        t.start()
        self._threads = [t]

    _TMP = ".writing"

    def finish(self, voxels, ori):
        import ctypes
        import threading

        import time

        t_0 = time.perf_counter()
        for t in self._threads:
            t.join()
        self.join_ms = (time.perf_counter() - t_0) * 1e3      # > 0: the background pre-fault had not finished
        from . import timing

        if timing.ENABLED:
            print("[mh-timing]   (.mat: waited %.1f ms for the background pre-fault)" % self.join_ms, file=sys.stderr, flush=True)
        if self._fatal:
            self.abort()
            raise self._fatal[0]
        X, Y, Z = self.X, self.Y, self.Z
        v = np.ascontiguousarray(np.asarray(voxels).reshape(-1, 3), dtype=np.int64)
        o = np.asarray(ori).reshape(-1, 3)
        if o.dtype != np.float32:                    # (float32 orientations convert exactly; anything else goes as float64)
            o = o.astype(np.float64)
        o = np.ascontiguousarray(o)
        jobs = ((self._files[0], None), (self._files[1], o))

        err, stored = [], [threading.Event(), threading.Event()]

        def store(k, h, val):
            # the caller waits for the ELEMENTS (then the files read back complete: a shared mapping and the page cache are
            # coherent); unmapping 400 MB of dirty pages and closing (1-2 ms) finish behind its back
            try:
                _lib.check(self._L.mh_mat_sparse_store_voxels(
                    h, v.ctypes.data_as(ctypes.c_void_p), None if val is None else val.ctypes.data_as(ctypes.c_void_p),
                    int(val is not None and val.dtype == np.float64), len(v), X, Y, Z), "mh_mat_sparse_store_voxels")
            except BaseException as e:
                err.append(e)
            stored[k].set()
            self._L.mh_mat_sparse_close(h)

        names = list(self._names)
        self._files = []
        self._closers = [threading.Thread(target=store, args=(k,) + jobs[k]) for k in (0, 1)]   # ctypes releases the GIL
        for t in self._closers:
            t.start()
        for e in stored:
            e.wait()
        # two-phase: the pair is renamed into place only when BOTH files hold their elements (an Occ3D.mat / Ori3D.mat pair
        # from two different runs is worse than none); a failed store leaves no temporary file behind
        if err:
            for n in names:
                try:
                    os.remove(n + self._TMP)
                except OSError:
                    pass
            raise err[0]
        for n in names:
            os.replace(n + self._TMP, n)

    def wait_closed(self):
        for t in getattr(self, "_closers", []):
            t.join()

    def abort(self):
        for t in self._threads:
            t.join()
        for k, h in enumerate(self._files):
            if h is not None:
                self._L.mh_mat_sparse_close(h)
                try:
                    os.remove(self._names[k] + self._TMP)
                except OSError:
                    pass
        self._files = []

    def __del__(self):          # refine left early (an exception): the mappings do not outlive the object
        try:
            self.abort()
        except Exception:
            pass


def merge_inner_points(voxels, ori, coarse_data, unvisible_index, voxel_min=VOXEL_MIN, voxel_size=VOXEL_SIZE,
                       grid_resolution=GRID_RESOLUTION):
    """The infer_inner merge of refine (PMVO.py:733-751): rows of DeepMVSHair's ours/raw.npy (N x 7: xyz, orientation,
    occupancy) that no view sees (`unvisible_index`, PMVO.compute_unvisible_points) overwrite the fitted volume -- sign
    canonicalised (ori.y > 0 -> negated, :739-740), quantised by p2v on a copy (:746), appended BEHIND the fitted voxels
    so that the "later rows win" rule of the writers reproduces the reference's fancy assignments (:747-748, where a
    later raw row also overwrites an earlier one in the same voxel).
    -> (voxels [G+M,3] int64, ori [G+M,3], un_visible_points [M,3] f32, unvisible_ori [M,3] f32); the last two are what
    the reference saves as coarse.npy / coarse_ori.npy (:749-750)."""
    cpoints = coarse_data[:, :3].astype(np.float32)
    coarse_ori = coarse_data[:, 3:6].astype(np.float32)
    coarse_ori[coarse_ori[:, 1] > 0] *= -1
    sel = np.asarray(unvisible_index, dtype=bool)
    un_visible_points, unvisible_ori = cpoints[sel], coarse_ori[sel]
    x, y, z = p2v(un_visible_points.copy(), np.asarray(voxel_min), voxel_size, grid_resolution)
    vox = np.concatenate([np.asarray(voxels, dtype=np.int64).reshape(-1, 3), np.stack([x, y, z], 1).astype(np.int64)])
    vori = np.concatenate([np.asarray(ori).reshape(-1, 3), unvisible_ori])
    return vox, vori, un_visible_points, unvisible_ori


def dense_from_sparse(grid_resolution, voxels, ori):
    """occ [X,Y,Z], ori [X,Y,Z,3] float64 as the reference holds them before saving (PMVO.py:698-726)."""
    g = tuple(int(v) for v in grid_resolution)
    occ, dense = np.zeros(g), np.zeros(g + (3,))
    v = np.asarray(voxels, dtype=np.int64).reshape(-1, 3)
    occ[v[:, 0], v[:, 1], v[:, 2]] = 1
    dense[v[:, 0], v[:, 1], v[:, 2]] = np.asarray(ori, dtype=np.float64).reshape(-1, 3)
    return occ, dense


def get_ground_truth_3D_occ(d, flip=False):
    """PMVO_utils.py:86-95 -> [Z,Y,X,1] float32."""
    import scipy.io

    occ = scipy.io.loadmat(d, verify_compressed_data_integrity=False)["Occ"].astype(np.float32)
    occ = np.expand_dims(np.transpose(occ, [2, 0, 1]), -1)
    if flip:
        occ = occ[:, :, ::-1, :]
    return np.ascontiguousarray(occ)


def get_ground_truth_3D_ori(d, flip=False, growInv=False):
    """PMVO_utils.py:98-113 -> [Z,Y,X,3] float32."""
    import scipy.io

    ori = scipy.io.loadmat(d, verify_compressed_data_integrity=False)["Ori"].astype(np.float32)
    ori = np.reshape(ori, [ori.shape[0], ori.shape[1], 3, -1])
    ori = ori.transpose([0, 1, 3, 2]).transpose(2, 0, 1, 3)
    if flip:
        ori = ori[:, :, ::-1, :] * np.array([-1.0, 1.0, 1.0])
    return np.ascontiguousarray(ori)


def voxel_to_points(voxels):
    """PMVO_utils.py:407-412."""
    voxel_min = torch.tensor([-0.32, -0.32, -0.24], dtype=torch.float, device=voxels.device)
    points = voxels * VOXEL_SIZE + voxel_min
    points[..., 1:] *= -1
    return points


def points_to_voxel(points):
    """PMVO_utils.py:414-420 (mutates its argument like the reference)."""
    voxel_min = torch.tensor([-0.32, -0.32, -0.24], dtype=torch.float, device=points.device)
    points[..., 1:] *= -1
    return (points - voxel_min) / VOXEL_SIZE


# ----------------------------------------------------------------------------------------- .hair strands
def load_strand(file):
    """.hair layout (PMVO_utils.py:47-66): u32 n_strands, u32 n_points, u16 count[n_strands], f32 xyz[...]."""
    with open(file, "rb") as f:
        (num_strand,) = struct.unpack("<I", f.read(4))
        (_point_count,) = struct.unpack("<I", f.read(4))
        segments = list(np.frombuffer(f.read(2 * num_strand), dtype="<u2").astype(int))
        num_points = int(sum(segments))
        points = np.frombuffer(f.read(4 * num_points * 3), dtype="<f4").astype(np.float64).reshape(-1, 3)
    return segments, points


def write_strand(points, path, segments):
    """PMVO_utils.py:69-83."""
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(segments)))
        f.write(struct.pack("<I", int(sum(segments))))
        f.write(np.asarray(segments, dtype="<u2").tobytes())
        f.write(np.asarray(points, dtype="<f4").reshape(-1, 3).tobytes())


def save_hair_strands(path, strands, bust_to_origin, translate=True):
    """PMVO_utils.py:662-680."""
    segments = [s.shape[0] for s in strands]
    points = np.concatenate(strands, 0)
    if translate:
        points = points - bust_to_origin
    write_strand(points, path, segments)
