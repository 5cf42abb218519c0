"""class PMVO -- the host-side mirror of the reference's `PMVO.py::class PMVO` (lines 13-529) on top of
the HIP library (include/mh_pmvo.h).  Same constructor, same methods, same return shapes/dtypes; every
method is one or two C-ABI calls on torch-owned device buffers.  There is no CPU path in here: without a
GPU and libmhpmvo.so construction fails loudly.

State set by Compute_Visible_and_Ori (as in the reference, PMVO.py:369-376): self.visible [V,N],
self.Ori [V,N,2], self.Conf [V,N], self.mask [V,N], self.Ori_patch [V,N,P,2], self.Conf_patch [V,N,P].
"""
import ctypes
import warnings
import os

import numpy as np
import torch

from . import _lib
from .camera import CAM_STRIDE, camera_records
from .timing import stage


def depth_offsets(num_sample=90):
    """The depth offsets of sample_next_3d_pos (PMVO.py:274-278): three torch.arange pieces, concatenated,
    cut to num_sample.  Built with the same CPU torch calls, so the values are the reference's bits."""
    s1 = torch.arange(-0.005, -0.001, 0.004 / (num_sample / 4))
    s2 = torch.arange(-0.001, 0.001, 0.002 / (num_sample / 2))
    s3 = torch.arange(0.001, 0.005, 0.004 / (num_sample / 4))
    return torch.cat([s1, s2, s3], 0)[:num_sample].numpy().astype(np.float32)


class PMVO:
    NUM_SAMPLE = 90          # PMVO.py:263
    RANKS = range(0, 20, 2)  # PMVO.py:50

    def __init__(self, camera, depths, Ori, Conf, masks, device="cuda:0", image_size=[1120, 1992], patch_size=5,
                 visible_threshold=1, conf_threshold=0.4):
        """camera: dict view -> Camera (insertion order = view order); depths[k] [H,W,3], Ori[k] [H,W,2],
        Conf[k] [H,W], masks[k] [H,W,3] numpy arrays (reference PMVO.py:14-28)."""
        self._init_common(device, image_size, patch_size, visible_threshold, conf_threshold)
        self.camera_dict = camera
        self.camera_key = list(camera.keys())
        self.camera = [camera[k] for k in self.camera_key]
        recs = camera_records(camera)
        H, W = int(image_size[0]), int(image_size[1])
        self._alloc(len(self.camera_key), H, W)
        st = _lib.stream_ptr()
        for i, k in enumerate(self.camera_key):
            d = torch.from_numpy(np.ascontiguousarray(np.asarray(depths[k]))).to(self.device).type(torch.float)
            o = torch.from_numpy(np.ascontiguousarray(np.asarray(Ori[k]))).to(self.device).type(torch.float)
            c = torch.from_numpy(np.ascontiguousarray(np.asarray(Conf[k]))).to(self.device).type(torch.float)
            m = torch.from_numpy(np.ascontiguousarray(np.asarray(masks[k]))).to(self.device).type(torch.float)
            assert d.shape[:2] == (H, W) and o.shape == (H, W, 2) and c.shape == (H, W) and m.shape[:2] == (H, W), \
                "map shapes do not match image_size=[H,W]"
            ds = d.shape[2] if d.dim() == 3 else 1
            ms = m.shape[2] if m.dim() == 3 else 1
            self._set_view(i, recs[i], d, ds, o, c, m, ms, st)
        torch.cuda.synchronize(self.device)

    @classmethod
    def from_planes(cls, cam_records, depth, ori, conf, mask, device="cuda:0", patch_size=5, visible_threshold=1,
                    conf_threshold=0.4, camera=None):
        """Device-resident compact planes: depth/conf/mask [V,H,W], ori [V,H,W,2] float32 tensors on `device`
        (what an on-GPU producer such as the Gabor stage hands over) + [V,48] camera records."""
        self = cls.__new__(cls)
        V, H, W = depth.shape
        self._init_common(device, [H, W], patch_size, visible_threshold, conf_threshold)
        self.camera_dict = camera
        self.camera_key = list(camera.keys()) if camera is not None else ["view_%03d" % i for i in range(V)]
        self.camera = list(camera.values()) if camera is not None else None
        recs = np.ascontiguousarray(cam_records, dtype=np.float32)
        assert recs.shape == (V, CAM_STRIDE)
        self._alloc(V, H, W)
        st = _lib.stream_ptr()
        for i in range(V):
            self._set_view(i, recs[i], depth[i].contiguous(), 1, ori[i].contiguous(), conf[i].contiguous(),
                           mask[i].contiguous(), 1, st)
        torch.cuda.synchronize(self.device)
        return self

    @classmethod
    def from_u8(cls, camera, depths, ori_u8, conf_u8, mask_u8, device="cuda:0", image_size=None, patch_size=5,
                visible_threshold=1, conf_threshold=0.4, lut=None, records=None):
        """The constructor for maps kept as their 8-bit pixel codes (pmvo_utils.load_maps_u8 or a maps pack):
        dicts view -> uint8 [H,W] for orientation / confidence / mask (or [V,H,W] arrays in view order), depth
        float32 [H,W] or [H,W,3]; numpy arrays or torch tensors (device tensors are used in place).  Decoded on the GPU through the 256-entry table of the loaders
        (pmvo_utils.map_code_lut), so the resident records equal those of PMVO(camera, <decoded float maps>).  The two codes of
        a pixel (orientation, confidence) stay resident as well, 2 B per pixel: the fused front end of forward() gathers a
        patch tap as those 2 bytes instead of a 16-byte record (option "tap_codes", default on; same results)."""
        from .pmvo_utils import map_code_lut

        self = cls.__new__(cls)
        keys = list(camera.keys())

        def get(maps, i):
            return maps[keys[i]] if isinstance(maps, dict) else maps[i]

        H, W = (int(image_size[0]), int(image_size[1])) if image_size is not None else get(ori_u8, 0).shape[:2]
        self._init_common(device, [H, W], patch_size, visible_threshold, conf_threshold)
        self.camera_dict = camera
        self.camera_key = keys
        self.camera = [camera[k] for k in keys]
        recs = camera_records(camera) if records is None else records      # (records: [V,48] override, parity tests)
        lut = np.ascontiguousarray(map_code_lut() if lut is None else lut, dtype=np.float32)
        assert lut.shape == (256, 4)
        self._alloc(len(keys), H, W)
        st = _lib.stream_ptr()
        def up(x, dtype=None):      # read-only memory maps are fine here: the tensor is only the source of a copy
            if torch.is_tensor(x):  # already on a device (Gabor stage / depth rasteriser output): no host round trip
                x = x.to(self.device)
                return (x if dtype is None else x.to(torch.float32)).contiguous()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                return torch.from_numpy(np.ascontiguousarray(x, dtype=dtype)).to(self.device)

        for i in range(len(keys)):
            d = up(get(depths, i), np.float32)
            planes = [up(get(m, i)) for m in (ori_u8, conf_u8, mask_u8)]
            assert d.shape[:2] == (H, W) and all(p.dtype == torch.uint8 and p.shape == (H, W) for p in planes), \
                "map shapes/dtypes do not match image_size=[H,W] / uint8"
            rec = np.ascontiguousarray(recs[i], dtype=np.float32)
            _lib.check(self._L.mh_ctx_set_view_u8(self._ctx, i, rec.ctypes.data_as(ctypes.c_void_p), _lib.ptr(d),
                                                  d.shape[2] if d.dim() == 3 else 1, _lib.ptr(planes[0]),
                                                  _lib.ptr(planes[1]), _lib.ptr(planes[2]),
                                                  lut.ctypes.data_as(ctypes.c_void_p), st), "mh_ctx_set_view_u8")
            torch.cuda.current_stream().synchronize()      # d / planes are read asynchronously by the pack kernel
        return self

    # ------------------------------------------------------------------ plumbing
    def _init_common(self, device, image_size, patch_size, visible_threshold, conf_threshold):
        if not torch.cuda.is_available():
            raise _lib.MhError("monohair_amd.PMVO needs a ROCm GPU: the hot path has no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MhError("monohair_amd.PMVO runs on HIP devices only (got %r)" % (device,))
        torch.cuda.set_device(self.device)
        self.image_size = [int(image_size[0]), int(image_size[1])]
        self.patch_size = int(patch_size)
        # taps run over range(-(size//2), size//2+1) in both directions (PMVO.py:494-495): an even size s gives the same
        # (s+1) x (s+1) window as s+1; the kernels are instantiated for the odd side
        self._side = 2 * (self.patch_size // 2) + 1
        self.visible_threshold = visible_threshold
        self.conf_threshold = conf_threshold
        self._stage = {}            # pinned upload rings of _upload_points, per launch stream
        self._scratch_need = {}     # N -> bytes of the search scratch (constant per context)
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        _lib.check(self._L.mh_ctx_create(self.device.index or 0, ctypes.byref(h)), "mh_ctx_create")
        self._ctx = h
        offs = depth_offsets(self.NUM_SAMPLE)
        _lib.check(self._L.mh_ctx_set_depth_offsets(self._ctx, offs.ctypes.data_as(ctypes.c_void_p), len(offs)),
                   "mh_ctx_set_depth_offsets")
        self._scratch = None
        self.bust_tree = self.scalp_tree = self.scalp_max = None
        self.base_view_override = None

    def _alloc(self, V, H, W):
        self.num_view = V
        _lib.check(self._L.mh_ctx_alloc_views(self._ctx, V, H, W), "mh_ctx_alloc_views")

    def _set_view(self, i, rec, d, ds, o, c, m, ms, st):
        rec = np.ascontiguousarray(rec, dtype=np.float32)
        _lib.check(self._L.mh_ctx_set_view(self._ctx, i, rec.ctypes.data_as(ctypes.c_void_p), _lib.ptr(d), ds,
                                           _lib.ptr(o), _lib.ptr(c), _lib.ptr(m), ms, st), "mh_ctx_set_view")
        # the pack kernel reads d/o/c/m asynchronously: keep them alive until it has run
        torch.cuda.current_stream().synchronize()

    def __del__(self):
        # an object dropped with uploads in flight must not release its pinned slabs under them (see _upload_points)
        try:
            for ring in getattr(self, "_stage", {}).values():
                self._drain_ring(ring)
            stg = getattr(self, "_stage_all", None)
            if stg is not None and stg[1] is not None:
                stg[1].synchronize()
            for ent in getattr(self, "_stage_named", {}).values():
                ent[1].synchronize()
        except Exception:
            pass
        try:
            if getattr(self, "_ctx", None):
                self._L.mh_ctx_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    LAB_OPTIONS = ("search_variant", "search_body", "tap_plane", "tap_codes", "taps_tile", "filter_rows")

    def set_option(self, key, value):
        """A supported option of the context (include/mh_pmvo.h: reproject_rule, reproject_fma_min_cols, sum_block, topk_order,
        gabor_variant, tap_plane_max_mb, line_rule, raster_subpixel_bits).  The lab switches of include/mh_pmvo_lab.h (A/B
        forms and cross-check kernels: LAB_OPTIONS) are routed to mh_ctx_set_lab_option so that tests and bench.py keep one call."""
        if key in self.LAB_OPTIONS:
            return self.set_lab_option(key, value)
        _lib.check(self._L.mh_ctx_set_option(self._ctx, key.encode(), int(value)), "mh_ctx_set_option")

    def set_lab_option(self, key, value):
        _lib.check(self._L.mh_ctx_set_lab_option(self._ctx, key.encode(), int(value)), "mh_ctx_set_lab_option")

    def set_head(self, bust_tree, scalp_tree, scalp_max):
        """The reference reads these from module globals (PMVO.py:99-106, 814-820)."""
        self.bust_tree, self.scalp_tree, self.scalp_max = bust_tree, scalp_tree, scalp_max

    def _dev_points(self, points):
        """[N,3] points -> float32 on the device.  numpy input is cast on the host first, as the reference does
        (`torch.from_numpy(points).type(torch.float).to(device)`, PMVO.py:40): same rounding, half the H2D bytes and no
        cast kernel on the stream."""
        if isinstance(points, np.ndarray):
            return self._upload_points(points)
        return points.to(self.device).type(torch.float).contiguous()

    def upload_all_points(self, points, head=0, after_head=None):
        """One asynchronous upload of a whole [M,3] host array (float32 cast on the host, PMVO.py:40) through one pinned
        staging buffer kept on the object; the drivers then hand device slices to forward().
        head > 0: rows [0, head) are converted and copied first, `after_head(dev)` is called (the driver launches the first
        chunks there), then the rest is converted and copied -- the host-side conversion of a large float64 array (1 ms per
        300 k points) then runs while the GPU already works.  Returns (device tensor, float32 host view of the staged rows --
        valid until the next call)."""
        m = int(points.shape[0])
        if m == 0:
            return torch.empty((0, 3), dtype=torch.float32, device=self.device), np.zeros((0, 3), np.float32)
        stg = getattr(self, "_stage_all", None)
        if stg is not None and stg[1] is not None:
            stg[1].synchronize()
        if stg is None or stg[0].shape[0] < m:
            stg = self._stage_all = [torch.empty((m, 3), dtype=torch.float32, pin_memory=True), torch.cuda.Event()]
        host = stg[0].numpy()[:m]
        dev = torch.empty((m, 3), dtype=torch.float32, device=self.device)
        cs = torch.cuda.current_stream(self.device)
        head = min(max(int(head), 0), m) if after_head is not None else 0
        for part, (lo, hi) in enumerate(((0, head), (head, m))):
            if hi > lo:
                np.copyto(host[lo:hi], points[lo:hi], casting="same_kind")
                _lib.check(self._L.mh_upload_async(self._ctx, stg[0].data_ptr() + lo * 12, dev.data_ptr() + lo * 12,
                                                   (hi - lo) * 12, cs.cuda_stream), "mh_upload_async")
            if part == 0 and after_head is not None:
                after_head(dev)
        stg[1].record(cs)
        return dev, host

    def _upload_points(self, points, stream=None):
        """Host numpy [N,3] -> float32 device tensor through a ring of PINNED staging slots per launch stream and
        an asynchronous copy.  `tensor.to(device)` from pageable memory blocks the host for ~0.2 ms per call (staging +
        wait) -- as long as a whole iteration takes on 8-bit maps, which made the loop of `optimize` host-bound there.
        The float64 -> float32 cast happens in the copy into the staging buffer (numpy's rounding = the reference's
        `.type(torch.float)`, PMVO.py:40)."""
        n = int(points.shape[0])
        if n == 0:
            return torch.empty((0, 3), dtype=torch.float32, device=self.device)
        cs = torch.cuda.current_stream(self.device) if stream is None else stream
        ring = self._stage.get(cs.cuda_stream)
        if ring is None or ring["cap"] < n:
            # ONE pinned slab of 32 slots per launch stream (3 MB), allocated on the stream's first call (a ring that grew
            # slot by slot paid a pinned allocation -- milliseconds, and it drains the device -- up to 32 times per stream
            # during the first few hundred iterations of a loop)
            # The copies go through mh_upload_pinned, which torch's host caching allocator does not track: before the old slab
            # is dropped (its pinned block could be handed out again at once), every copy still queued out of it must be over.
            self._drain_ring(ring)
            cap = max(n, 8192)
            slab = torch.empty((self._STAGE_SLOTS, cap, 3), dtype=torch.float32, pin_memory=True)
            ring = self._stage[cs.cuda_stream] = {"cap": cap, "slab": slab, "np": slab.numpy(), "i": 0,
                                                  "ptr": slab.data_ptr(), "ev": [None] * self._STAGE_SLOTS}
        # slots are used in ring order (copies of one stream complete in order); a slot whose copy is still pending means the
        # host is 32 iterations ahead of a GPU-bound loop: only then does it wait
        k = ring["i"] % self._STAGE_SLOTS
        ring["i"] += 1
        ev = ring["ev"][k]
        if ev is None:
            ev = ring["ev"][k] = torch.cuda.Event()
        elif not ev.query():
            ev.synchronize()
        np.copyto(ring["np"][k, :n], points, casting="same_kind")
        dev = torch.empty((n, 3), dtype=torch.float32, device=self.device)
        # (not `dev.copy_(pinned, non_blocking=True)`: that also records an allocator-tracking event per call -- 70 -> 55 us of
        # host time per forward())
        _lib.check(self._L.mh_upload_pinned(self._ctx, ring["ptr"] + k * ring["cap"] * 12, dev.data_ptr(), n * 12,
                                            cs.cuda_stream), "mh_upload_pinned")
        ev.record(cs)
        return dev

    _STAGE_SLOTS = 32

    @staticmethod
    def _drain_ring(ring):
        """wait for every asynchronous copy that still reads from a staging ring's pinned slab"""
        for ev in (ring or {}).get("ev", ()):
            if ev is not None:
                ev.synchronize()

    # ------------------------------------------------------------------ reference methods
    def Compute_Visible_and_Ori(self, points):
        """PMVO.py:346-376 -> one mh_project_gather launch."""
        points = self._dev_points(points)
        V, N, P = self.num_view, points.shape[0], self._side ** 2
        f = dict(dtype=torch.float32, device=self.device)
        self.visible = torch.empty((V, N), **f)
        self.Ori = torch.empty((V, N, 2), **f)
        self.Conf = torch.empty((V, N), **f)
        self.mask = torch.empty((V, N), **f)
        self._Ori_patch = torch.empty((V, N, P, 2), **f)
        self._Conf_patch = torch.empty((V, N, P), **f)
        self._pixf = torch.empty((V, N, 2), **f)
        self._points = points
        _lib.check(self._L.mh_project_gather(self._ctx, _lib.ptr(points), N, self._side, _lib.ptr(self.visible),
                                             _lib.ptr(self.Ori), _lib.ptr(self.Conf), _lib.ptr(self.mask),
                                             _lib.ptr(self._Ori_patch), _lib.ptr(self._Conf_patch),
                                             _lib.ptr(self._pixf), _lib.stream_ptr()), "mh_project_gather")

    def _materialise_patches(self):
        """forward() does not write the [V,N,P,..] patch tensors (its fused front end builds the search's tap
        lists straight from the maps); they are produced on first access so the attribute surface of the
        reference (PMVO.py:374-376) is kept."""
        if getattr(self, "_Ori_patch", None) is None and getattr(self, "_points", None) is not None:
            self.Compute_Visible_and_Ori(self._points)

    @property
    def Ori_patch(self):
        self._materialise_patches()
        return self._Ori_patch

    @property
    def Conf_patch(self):
        self._materialise_patches()
        return self._Conf_patch

    def _topk32(self, stp=None):
        V, N = self.visible.shape
        idx = torch.empty((20, N), dtype=torch.int32, device=self.device)
        val = torch.empty((20, N), dtype=torch.float32, device=self.device)
        _lib.check(self._L.mh_topk_views(self._ctx, _lib.ptr(self.visible), _lib.ptr(self.Conf), N, _lib.ptr(idx),
                                         _lib.ptr(val), _lib.stream_ptr() if stp is None else stp), "mh_topk_views")
        return idx, val

    def Find_max_conf_from_visible_view(self):
        """PMVO.py:339-343 -> (base_view_index [20,N] int64, base_view_conf [20,N])."""
        idx, val = self._topk32()
        return idx.long(), val

    # ------------------------------------------------------------------ the reference's intermediate methods
    # (PMVO.forward above is fused and does not call them; same names, arguments and returns as PMVO.py)
    def _view_index(self, view):
        """camera key / Camera object / integer -> index of the resident view"""
        if isinstance(view, (int, np.integer)):
            return int(view)
        if isinstance(view, str):
            return self.camera_key.index(view)
        for i, c in enumerate(self.camera or []):
            if c is view:
                return i
        name = getattr(view, "id", None)
        if name in self.camera_key:
            return self.camera_key.index(name)
        raise _lib.MhError("unknown view %r" % (view,))

    def project_points(self, points, camera, image_size=None):
        """PMVO.py:378-397 -> (uv [N,2] long as (row, col), z' = -z/2 [N], out-of-image flags [N]).  `camera`: one of
        the resident cameras (object, key or index); image_size must be the resident one."""
        pts = self._dev_points(points)
        N = pts.shape[0]
        if image_size is not None and [int(image_size[0]), int(image_size[1])] != self.image_size:
            raise _lib.MhError("project_points: image_size differs from the resident maps")
        rc = torch.empty((N, 2), dtype=torch.int32, device=self.device)
        zp = torch.empty((N,), dtype=torch.float32, device=self.device)
        oob = torch.empty((N,), dtype=torch.bool, device=self.device)
        _lib.check(self._L.mh_project_points(self._ctx, self._view_index(camera), _lib.ptr(pts), N, _lib.ptr(rc),
                                             _lib.ptr(zp), _lib.ptr(oob), None, _lib.stream_ptr()), "mh_project_points")
        return rc.long(), zp, oob

    def _gather(self, uv, view, size=1, want_mask=False):
        uv = torch.as_tensor(uv).to(self.device).long().contiguous()
        size = 2 * (int(size) // 2) + 1          # range(-(size // 2), size // 2 + 1): an even size is the next odd window
        N, P = uv.shape[0], size * size
        rec = torch.empty((N, P, 4), dtype=torch.float32, device=self.device)
        mask = torch.empty((N, P), dtype=torch.float32, device=self.device) if want_mask else None
        _lib.check(self._L.mh_gather_pixels(self._ctx, self._view_index(view), _lib.ptr(uv), N, size, _lib.ptr(rec),
                                            _lib.ptr(mask), _lib.stream_ptr()), "mh_gather_pixels")
        return rec, mask

    def get_depth(self, uv, view):
        """PMVO.py:482-485"""
        return self._gather(uv, view)[0][:, 0, 3]

    def get_ori(self, uv, view):
        """PMVO.py:487-489"""
        return self._gather(uv, view)[0][:, 0, 0:2]

    def get_conf(self, uv, view):
        """PMVO.py:517-519 (raw, unclamped)"""
        return self._gather(uv, view)[0][:, 0, 2]

    def get_mask(self, uv, view):
        """PMVO.py:521-523"""
        return self._gather(uv, view, want_mask=True)[1][:, 0]

    def get_ori_patch(self, uv, view, size=1):
        """PMVO.py:491-502 -> [N, side*side, 2], side = 2 * (size // 2) + 1"""
        return self._gather(uv, view, size)[0][..., 0:2].contiguous()

    def get_c_patch(self, uv, view, size=1):
        """PMVO.py:504-515 -> [N, side*side], side = 2 * (size // 2) + 1"""
        return self._gather(uv, view, size)[0][..., 2].contiguous()

    def compute_visible(self, depth, z):
        """PMVO.py:525-529 (z is already -z_cam/2*255)"""
        depth = torch.as_tensor(depth).to(self.device).float().contiguous()
        z = torch.as_tensor(z).to(self.device).float().contiguous().expand_as(depth).contiguous()
        out = torch.empty_like(depth)
        _lib.check(self._L.mh_compute_visible(self._ctx, _lib.ptr(depth), _lib.ptr(z), depth.numel(), _lib.ptr(out),
                                              _lib.stream_ptr()), "mh_compute_visible")
        return out

    def compute_weight(self, visible, Conf, mask):
        """PMVO.py:211-215: (visible != -1) * Conf; the mask `where` of the reference is an identity"""
        return torch.where(visible == -1, torch.zeros_like(visible), torch.ones_like(visible)) * Conf

    def sample_next_3d_pos(self, points, base_view_index, num_sample=90):
        """PMVO.py:263-335 -> (sample_points [N,num_sample,3], surface_points [N,3]); needs Compute_Visible_and_Ori
        on the same points.  (The reference's surface-point assignment at :333-334 writes into a temporary, so
        surface_points is a copy of points there too.)"""
        pts = self._dev_points(points)
        N = pts.shape[0]
        if getattr(self, "Ori", None) is None or self.Ori.shape[1] != N:
            raise _lib.MhError("sample_next_3d_pos: call Compute_Visible_and_Ori(points) first")
        base = torch.as_tensor(base_view_index).to(self.device).to(torch.int32).contiguous()
        offs = torch.from_numpy(depth_offsets(num_sample)).to(self.device)
        out = torch.empty((N, offs.shape[0], 3), dtype=torch.float32, device=self.device)
        _lib.check(self._L.mh_sample_next(self._ctx, _lib.ptr(pts), _lib.ptr(base), _lib.ptr(self.Ori), _lib.ptr(offs), N,
                                          offs.shape[0], _lib.ptr(out), _lib.stream_ptr()), "mh_sample_next")
        return out, pts.clone()

    def compute_reproject_ori(self, points, sample_next_points):
        """PMVO.py:219-241 -> [V, N, num_sample, 2]"""
        pts = self._dev_points(points)
        smp = torch.as_tensor(sample_next_points).to(self.device).float().contiguous()
        N, S = smp.shape[0], smp.shape[1]
        D = torch.empty((self.num_view, N, S, 2), dtype=torch.float32, device=self.device)
        _lib.check(self._L.mh_reproject_ori(self._ctx, _lib.ptr(pts), _lib.ptr(smp), N, S, _lib.ptr(D),
                                            _lib.stream_ptr()), "mh_reproject_ori")
        return D

    def compute_points_prj_ori(self, points, next_points):
        """PMVO.py:243-260 -> [V, N, 2]"""
        nxt = torch.as_tensor(next_points).to(self.device).float()
        return self.compute_reproject_ori(points, nxt[:, None, :])[:, :, 0, :].contiguous()

    def compute_prj_loss(self, Prj_Ori_2D, Ori=None, weight=None):
        """PMVO.py:151-209 -> (min_loss [N], min_index [N] long, high_conf_index [N] bool) on the patches of the last
        Compute_Visible_and_Ori (the reference ignores its `Ori` and `weight` arguments too)."""
        D = torch.as_tensor(Prj_Ori_2D).to(self.device).float().contiguous()
        V, N, S, _ = D.shape
        P = self._side ** 2
        loss = torch.empty((N,), dtype=torch.float32, device=self.device)
        idx = torch.empty((N,), dtype=torch.int64, device=self.device)
        hc = torch.empty((N,), dtype=torch.bool, device=self.device)
        _lib.check(self._L.mh_prj_loss(self._ctx, _lib.ptr(D), _lib.ptr(self.Ori_patch), _lib.ptr(self.Conf_patch),
                                       _lib.ptr(self.visible), V, N, S, P, float(self.conf_threshold), _lib.ptr(loss),
                                       _lib.ptr(idx), _lib.ptr(hc), None, _lib.stream_ptr()), "mh_prj_loss")
        return loss, idx, hc

    def side_streams(self, n=3):
        """n HIP streams owned by this object (created once; the per-stream search scratch is keyed by them)."""
        if len(getattr(self, "_side_streams", [])) < n:
            self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(n)]
        return self._side_streams[:n]

    def aux_stream(self, name, priority=None):
        """A named HIP stream owned by this object, created on first use (the drivers' copy / side / prefetch streams)."""
        d = self.__dict__.setdefault("_aux_streams", {})
        if name not in d:
            d[name] = (torch.cuda.Stream(device=self.device) if priority is None
                       else torch.cuda.Stream(device=self.device, priority=int(priority)))
        return d[name]

    _TORCH_OF = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8}

    def stage_fill(self, key, arr, dtype=np.float32):
        """First half of stage_upload, safe to run on a worker thread (one key per thread): convert / copy `arr` into the
        pinned staging buffer of `key` (waiting for the previous copy out of it).  -> handle for stage_issue."""
        a = np.asarray(arr)
        dt = np.dtype(dtype)
        nbytes = a.size * dt.itemsize
        if nbytes == 0:
            return (None, a.shape, dt, 0)
        pool = self.__dict__.setdefault("_stage_named", {})
        ent = pool.get(key)
        if ent is not None:
            ent[1].synchronize()          # the previous copy out of this buffer
        if ent is None or ent[0].numel() < nbytes:
            ent = pool[key] = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True), torch.cuda.Event()]
        host = ent[0].numpy()[:nbytes].view(dt).reshape(a.shape)
        np.copyto(host, a, casting="same_kind")
        return (ent, a.shape, dt, nbytes)

    def stage_issue(self, handle):
        """Second half: the asynchronous copy of a filled staging buffer on the current stream -> fresh device tensor."""
        ent, shape, dt, nbytes = handle
        dev = torch.empty(shape, dtype=self._TORCH_OF[dt], device=self.device)
        if nbytes == 0:
            return dev
        cs = torch.cuda.current_stream(self.device)
        _lib.check(self._L.mh_upload_pinned(self._ctx, ent[0].data_ptr(), dev.data_ptr(), nbytes, cs.cuda_stream),
                   "mh_upload_pinned")
        ent[1].record(cs)
        return dev

    def stage_upload(self, key, arr, dtype=np.float32):
        """Host array -> fresh device tensor of `dtype` through a pinned staging buffer kept per `key`: one conversion /
        copy on the host (numpy's rounding for float64 -> float32 = the reference's `.type(torch.float)`, PMVO.py:40), one
        asynchronous copy on the current stream.  `tensor.to(device)` from pageable memory blocks the host for the whole
        transfer; the drivers upload a pass's arrays while the GPU works."""
        return self.stage_issue(self.stage_fill(key, arr, dtype))

    def start_refine_prefetch(self, pts_dev, sub_num=5000, k=100):
        """What refine() needs of the POINTS alone (PMVO.py:605-612 the 100 nearest neighbours of every point; :96-137 the
        head-filter votes and the scalp test) is prepared on a stream of its own while optimize()'s iterations run: see
        RefinePrefetch.  Called by optimize() once its launches are queued; refine() adopts the result if it is handed
        the same points (compared on the device, bit for bit)."""
        if os.environ.get("MH_REFINE_PREFETCH", "1") == "0":
            self._prefetch = None
            return None
        self._prefetch = RefinePrefetch(self, pts_dev, sub_num, k)
        return self._prefetch

    def take_refine_prefetch(self, pts_dev, sub_num, k):
        """The prefetch of start_refine_prefetch if it was made for exactly these points and thresholds, else None."""
        pf, self._prefetch = getattr(self, "_prefetch", None), None
        if pf is None:
            return None
        return pf.adopt(pts_dev, sub_num, k)

    def _get_scratch(self, N, key=None):
        """Tap-list scratch of the search, one buffer per launch stream (chunks of `optimize` are independent and
        may be in flight on different streams)."""
        need = self._scratch_need.get(N)
        if need is None:
            need = self._scratch_need[N] = int(self._L.mh_search_scratch_bytes(self._ctx, N, self._side))
        if key is None:
            key = torch.cuda.current_stream().cuda_stream
        if self._scratch is None:
            self._scratch = {}
        buf = self._scratch.get(key)
        if buf is None or buf.numel() < need:
            buf = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._scratch[key] = buf
        return buf, need

    def search_work(self, N, base_val=None):
        """What the last forward() on the current stream executed: (list lengths [V,N] int64, usable ranks [N] or None).
        The preparation kernels leave the per-(view, point) tap-list lengths in the scratch (0 = the view does not see
        the point); with base_val [20,N] the number of base-view ranks the search evaluated per point (rank 0 always,
        later ranks up to the last one with base_view_conf > 0, PMVO.py:64)."""
        buf, _ = self._get_scratch(N)
        off = int(self._L.mh_search_counts_offset(self._ctx, N, self._side))
        cnt = buf[off:off + self.num_view * N].view(self.num_view, N).to(torch.int64)
        nvalid = None
        if base_val is not None:
            pos = base_val[list(self.RANKS)] > 0                      # [10, N]
            last = (pos * torch.arange(1, pos.shape[0] + 1, device=pos.device)[:, None]).amax(0)
            nvalid = torch.clamp(last, min=1)
        return cnt, nvalid

    def forward(self, points, base_view=None, extras=False, fused=True, out=None):
        """PMVO.py:39-78.  points: numpy [N,3].  Returns (points, line_ori [N,3], min_loss [N],
        high_conf [N] bool) on the device.  base_view=(idx [20,N], val [20,N]) injects a base-view ranking
        (parity tests: torch.topk's tie order is unspecified).  fused=False runs Compute_Visible_and_Ori and
        the tap preparation as separate kernels through the materialised patch tensors (same results).
        out=(line_ori [N,3] float32, min_loss [N] float32, high_conf [N] bool): contiguous device tensors the search
        writes into (the driver `optimize` passes slices of one result buffer: no per-chunk concatenation)."""
        ranks = list(self.RANKS)
        f = dict(dtype=torch.float32, device=self.device)
        cs = torch.cuda.current_stream(self.device)          # looked up once: ~8 us of Python per call
        stp = ctypes.c_void_p(cs.cuda_stream)
        d_ = torch.Tensor.data_ptr          # raw addresses: ctypes converts ints to void* without a wrapper object each
        if fused:
            points = self._upload_points(points, cs) if isinstance(points, np.ndarray) else self._dev_points(points)
            V, N = self.num_view, points.shape[0]
            self.visible = torch.empty((V, N), **f)
            self.Ori = torch.empty((V, N, 2), **f)
            self.Conf = torch.empty((V, N), **f)
            self.mask = torch.empty((V, N), **f)
            self._Ori_patch = self._Conf_patch = self._pixf = None
            self._points = points
            scratch, need = self._get_scratch(N, cs.cuda_stream)
        else:
            self.Compute_Visible_and_Ori(points)
            points = self._points
            N = points.shape[0]
            scratch, need = self._get_scratch(N, cs.cuda_stream)
        if out is not None:
            line_ori, min_loss, hc = out
            assert line_ori.shape == (N, 3) and min_loss.shape == (N,) and hc.shape == (N,) and hc.dtype == torch.bool
            assert line_ori.is_contiguous() and min_loss.is_contiguous() and hc.is_contiguous() and line_ori.is_cuda
        else:
            line_ori = torch.empty((N, 3), **f)
            min_loss = torch.empty((N,), **f)
            hc = torch.empty((N,), dtype=torch.bool, device=self.device)      # the kernel writes 0/1 bytes
        bs = torch.empty((N, 3), **f) if extras else None
        br = torch.empty((N,), dtype=torch.int32, device=self.device) if extras else None
        bi = torch.empty((N,), dtype=torch.int32, device=self.device) if extras else None
        if fused and base_view is None:
            # the whole iteration in one call: projection / visibility / tap lists, base-view ranking, loss search
            bidx32 = torch.empty((20, N), dtype=torch.int32, device=self.device)
            bval = torch.empty((20, N), **f)
            if N:
                _lib.check(self._L.mh_forward(
                    self._ctx, d_(points), N, self._side, float(self.conf_threshold), len(ranks), ranks[1] - ranks[0],
                    d_(self.visible), d_(self.Ori), d_(self.Conf), d_(self.mask), d_(scratch), need, d_(bidx32), d_(bval),
                    d_(line_ori), d_(min_loss), d_(hc), _lib.ptr(bs), _lib.ptr(br), _lib.ptr(bi), stp), "mh_forward")
        else:
            if fused:
                _lib.check(self._L.mh_forward_prepare(self._ctx, _lib.ptr(points), N, self._side,
                                                      float(self.conf_threshold), _lib.ptr(self.visible),
                                                      _lib.ptr(self.Ori), _lib.ptr(self.Conf), _lib.ptr(self.mask),
                                                      _lib.ptr(scratch), need, stp), "mh_forward_prepare")
            if base_view is None:
                bidx32, bval = self._topk32(stp)       # int32 end to end: no int64 round trip on the hot path
            else:
                bidx32 = torch.as_tensor(base_view[0]).to(self.device).to(torch.int32).contiguous()
                bval = torch.as_tensor(base_view[1]).to(self.device).type(torch.float).contiguous()
            if fused:
                _lib.check(self._L.mh_search_prepared(
                    self._ctx, _lib.ptr(points), N, self._side, float(self.conf_threshold), len(ranks),
                    ranks[1] - ranks[0], _lib.ptr(self.Ori), _lib.ptr(bidx32), _lib.ptr(bval), _lib.ptr(scratch),
                    _lib.ptr(line_ori), _lib.ptr(min_loss), _lib.ptr(hc), _lib.ptr(bs), _lib.ptr(br), _lib.ptr(bi),
                    stp), "mh_search_prepared")
            else:
                _lib.check(self._L.mh_search_forward(
                    self._ctx, _lib.ptr(points), N, self._side, float(self.conf_threshold), len(ranks),
                    ranks[1] - ranks[0], _lib.ptr(self.visible), _lib.ptr(self.Ori), _lib.ptr(self._pixf),
                    _lib.ptr(self._Ori_patch), _lib.ptr(self._Conf_patch), _lib.ptr(bidx32), _lib.ptr(bval),
                    _lib.ptr(scratch), need, _lib.ptr(line_ori), _lib.ptr(min_loss), _lib.ptr(hc), _lib.ptr(bs),
                    _lib.ptr(br), _lib.ptr(bi), stp), "mh_search_forward")
        out = (points, line_ori, min_loss, hc)
        if extras:
            return out + (dict(best_sample=bs, best_rank=br, best_s=bi, base_idx=bidx32.long(), base_val=bval),)
        return out

    __call__ = forward

    def prj_loss_of(self, points, ori):
        """compute_reproject_ori + compute_prj_loss for next = points + ori*0.005/4 (PMVO.py:86-90) on the
        state of the last Compute_Visible_and_Ori: (loss [N], high_conf [N] bool)."""
        N = points.shape[0]
        ori = ori.to(self.device).type(torch.float).contiguous()
        loss = torch.empty((N,), dtype=torch.float32, device=self.device)
        hc = torch.empty((N,), dtype=torch.uint8, device=self.device)
        _lib.check(self._L.mh_refine_loss(self._ctx, _lib.ptr(points), _lib.ptr(ori), 0.005, 4.0, N,
                                          self._side, float(self.conf_threshold), _lib.ptr(self.visible),
                                          _lib.ptr(self.Ori_patch), _lib.ptr(self.Conf_patch), _lib.ptr(loss),
                                          _lib.ptr(hc), _lib.stream_ptr()), "mh_refine_loss")
        return loss, hc.bool()

    def refine(self, points, ori, head_top=None):
        """PMVO.py:81-93: loss of a given direction; -1 where filter_head_points fires.  head_top: optional
        precomputed host part of filter_head_points (head_top_mask) so that the call does not synchronise."""
        self.Compute_Visible_and_Ori(points)
        points = self._points
        filter_index = self.filter_head_points(points, self.visible_threshold, head_top=head_top)
        loss, _ = self.prj_loss_of(points, ori)
        return torch.where(filter_index, torch.full_like(loss, -1.0), loss)

    def head_top_mask_device(self, points_dev):
        """The scalp half of filter_head_points (PMVO.py:98-107) for float32 points on the device: within 4 cm of the
        scalp and more than 1 cm below its top -> uint8 (0/1) tensor on the device.  The reference asks a scipy KDTree for the
        nearest scalp vertex (float64); here the distance to the nearest vertex is found exhaustively on the GPU in
        float64 (mh_nearest_distance: the same sum of squares and one sqrt), so the 4 cm decision is scipy's."""
        if self.scalp_tree is None:
            raise _lib.MhError("filter_head_points needs set_head(bust_tree, scalp_tree, scalp_max)")
        if getattr(self, "_scalp_dev", None) is None or self._scalp_dev[0] is not self.scalp_tree:
            ref = np.ascontiguousarray(np.asarray(self.scalp_tree.data, dtype=np.float64).reshape(-1, 3))
            self._scalp_dev = (self.scalp_tree, torch.from_numpy(ref).to(self.device))
        ref = self._scalp_dev[1]
        pts = points_dev if (points_dev.dtype == torch.float32 and points_dev.is_contiguous()) else \
            points_dev.to(self.device).type(torch.float).contiguous()
        N = pts.shape[0]
        mask = torch.empty((N,), dtype=torch.uint8, device=self.device)
        # (the height test in float64, as numpy evaluates `float32 array < float64 scalar` where the goldens were made)
        _lib.check(self._L.mh_nearest_distance(self._ctx, _lib.ptr(pts), N, _lib.ptr(ref), ref.shape[0], None, 0.04,
                                               float(self.scalp_max[2] - 0.01), _lib.ptr(mask), _lib.stream_ptr()),
                   "mh_nearest_distance")
        return mask

    def head_top_mask(self, points_numpy):
        """numpy in / numpy out form of head_top_mask_device (float32 points, as the reference passes them)."""
        pts = np.ascontiguousarray(np.asarray(points_numpy, dtype=np.float32).reshape(-1, 3))
        return self.head_top_mask_device(torch.from_numpy(pts).to(self.device)).cpu().numpy().astype(bool)

    def replace_dissimilar(self, center, ori, threshold=0.95):
        """ori[n] <- center[n] where |cos(center, ori)| < threshold, in place on the device (PMVO.py:631-636)."""
        N = ori.shape[0]
        _lib.check(self._L.mh_replace_dissimilar(self._ctx, _lib.ptr(center), _lib.ptr(ori), float(threshold), N,
                                                 _lib.stream_ptr()), "mh_replace_dissimilar")

    def _votes(self, points, want, visible_threshold=None, raw=False):
        """mask / visibility votes of mh_filter_points for the requested outputs; raw=True returns the kernel's uint8 0/1
        arrays instead of bool tensors.  (The points of one call are one batch of the reference: its sums over views add the
        trailing len mod 32 points of a batch in another order, include/mh_pmvo.h.)"""
        points = self._dev_points(points)
        N = points.shape[0]
        bufs = [torch.empty((N,), dtype=torch.uint8, device=self.device) if w else None for w in want]
        vt = self.visible_threshold if visible_threshold is None else visible_threshold
        _lib.check(self._L.mh_filter_points(self._ctx, _lib.ptr(points), N, self._side,
                                            float(self.conf_threshold), float(vt), _lib.ptr(bufs[0]),
                                            _lib.ptr(bufs[1]), _lib.ptr(bufs[2]), _lib.ptr(bufs[3]), 0, 0, 0,
                                            _lib.stream_ptr()), "mh_filter_points")
        if raw:
            return points, bufs
        return points, [None if b is None else b.bool() for b in bufs]

    def filter_points(self, points):
        """PMVO.py:402-459 -> (surface_index, surface_points, filter_index)."""
        points, (surf, filt, _, _) = self._votes(points, (True, True, False, False))
        return surf, points[surf], filt

    def compute_unvisible_points(self, points):
        """PMVO.py:461-480."""
        _, (_, _, unv, _) = self._votes(points, (False, False, True, False))
        return unv

    def filter_head_points(self, points, visible_threshold, head_top=None):
        """PMVO.py:96-144: mask vote on the GPU; the scalp KDTree query (scipy, float64) stays on the host exactly
        as in the reference (the bust query's result is unused there, :99).  head_top: optional precomputed
        head_top_mask of the same points (bool tensor on the device)."""
        pts, (_, _, _, head) = self._votes(points, (False, False, False, True), visible_threshold)
        if head_top is None:
            head_top = self.head_top_mask_device(pts)
        return torch.logical_and(head, ~head_top.bool())


class RefinePrefetch:
    """The point-only inputs of refine()'s smoothing loop, prepared while optimize() runs (single rank).

    refine's neighbour table `KDTree(points).query(points, 100)` (PMVO.py:605-612), the head-filter votes of every point
    (:110-137, per 5000-point chunk) and the scalp test (:98-107) depend on the points alone, and the points exist before
    optimize() starts.  A helper thread builds the grid, runs the self-query (with its host-side retries) and the two
    vote kernels on a low-priority stream of its own; the search kernels of optimize() keep the GPU, these fill what they
    leave.  refine() adopts the result only if the points it is handed equal these bit for bit (mh_buffers_differ on the
    device) and the thresholds are the ones used here; otherwise it computes everything itself as before."""

    def __init__(self, pmvo, pts_dev, sub_num, k):
        import threading

        self.pmvo, self.pts_dev, self.sub_num, self.k = pmvo, pts_dev, int(sub_num), int(k)
        self.n = int(pts_dev.shape[0])
        self.conf_threshold, self.visible_threshold = float(pmvo.conf_threshold), float(pmvo.visible_threshold)
        self.scalp_tree = pmvo.scalp_tree
        self.error = None
        self.grid = self.index_all = self.head_all = self.head_top_all = None
        prio = os.environ.get("MH_PREFETCH_PRIORITY", "low")
        rng = (0, 0)
        try:
            rng = torch.cuda.Stream.priority_range()        # (least, greatest), e.g. (0, -1)
        except Exception:
            pass
        self.stream = pmvo.aux_stream("prefetch_" + prio, priority=(rng[0] if prio == "low" else 0))
        self.ready = torch.cuda.Event()
        self.ready.record(torch.cuda.current_stream(pmvo.device))      # pts_dev's upload is queued on the caller's stream
        self.done = torch.cuda.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def _run(self):
        from .pmvo_utils import GridKNN

        pm = self.pmvo
        try:
            with torch.cuda.device(pm.device), torch.cuda.stream(self.stream):
                self.stream.wait_event(self.ready)
                if self.n:
                    self.grid = GridKNN(self.pts_dev, k_hint=self.k, device=pm.device)
                    self.index_all = self.grid.query(self.pts_dev, self.k, int32=True, self_query=True).contiguous()
                    self.head_all = torch.empty((self.n,), dtype=torch.uint8, device=pm.device)
                    # (rows in the cell order of the grid the neighbour search just built: 64 spatial neighbours per wave)
                    _lib.check(pm._L.mh_filter_points_ordered(pm._ctx, _lib.ptr(self.pts_dev), self.n, pm._side,
                                                              self.conf_threshold, self.visible_threshold, None, None, None,
                                                              _lib.ptr(self.head_all), self.sub_num, 0, self.n,
                                                              _lib.ptr(self.grid.cell_order()), _lib.stream_ptr()),
                               "mh_filter_points_ordered")
                    if self.scalp_tree is not None:
                        self.head_top_all = pm.head_top_mask_device(self.pts_dev)
                self.done.record(self.stream)
        except BaseException as e:          # refine() then prepares these itself
            self.error = e

    def adopt(self, pts_dev, sub_num, k):
        """-> self (results valid on the CURRENT stream) or None.  Joins the helper thread."""
        self.thread.join()
        pm = self.pmvo
        if (self.error is not None or self.n != int(pts_dev.shape[0]) or self.n == 0 or self.sub_num != int(sub_num)
                or self.k != int(k) or self.conf_threshold != float(pm.conf_threshold)
                or self.visible_threshold != float(pm.visible_threshold) or self.scalp_tree is not pm.scalp_tree
                or self.head_top_all is None):
            return None
        cur = torch.cuda.current_stream(pm.device)
        cur.wait_event(self.done)
        flag = torch.empty(1, dtype=torch.int32, device=pm.device)
        _lib.check(pm._L.mh_buffers_differ(pm._ctx, _lib.ptr(pts_dev), _lib.ptr(self.pts_dev), self.n * 12, _lib.ptr(flag),
                                           _lib.stream_ptr()), "mh_buffers_differ")
        if int(flag.cpu().numpy()[0]) != 0:
            return None
        for t in (self.index_all, self.head_all, self.head_top_all):
            t.record_stream(cur)              # allocated on the prefetch stream, used on this one from now on
        self.grid.record_stream(cur)
        return self


# =====================================================================================================
# Drivers -- mirrors of the module-level functions of the reference's PMVO.py (:535-764).  Chunking, file
# names and dtypes are the reference's (they are its checkpoint/resume mechanism); with torch.distributed
# initialised the independent chunks are dealt round-robin to the ranks (monohair_amd.dist).
# =====================================================================================================
def _filter_single(points, pmvo, step, num_sub_p):
    """filter_negative_points on one rank: the candidates go up in a few large blocks (the float64 -> float32 conversion of
    block k+1 runs on the host while the GPU votes on block k), one vote launch per block -- the kernel is told where its
    points sit in the reference's N//30-sized pieces (PMVO.py:540-552; include/mh_pmvo.h: batch / row0 / total) -- and the
    surface points are compacted on the device (`covered[surface].astype(float32)` = the selected rows of the float32
    copy the votes were computed on).  -> (flags [total,2] uint8 numpy, surface_points float32 [S,3] numpy, pinned)."""
    from .pmvo_utils import spatial_order

    dev = pmvo.device
    L, ctx = pmvo._L, pmvo._ctx
    total = min(int(points.shape[0]), step * num_sub_p)
    per = max(1, -(-step // 4)) * num_sub_p                      # four blocks, cut at piece boundaries
    bounds = [(lo, min(lo + per, total)) for lo in range(0, total, per)]
    surf = torch.empty((total,), dtype=torch.uint8, device=dev)
    filt = torch.empty((total,), dtype=torch.uint8, device=dev)
    out = torch.empty((total, 3), dtype=torch.float32, device=dev)
    cnt = torch.zeros((len(bounds) + 1,), dtype=torch.int32, device=dev)
    scratch = torch.empty(int(L.mh_select_scratch_bytes(per)), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr()
    off = lambda t, row, width=1: ctypes.c_void_p(t.data_ptr() + row * width * t.element_size())   # noqa: E731
    for k, (lo, hi) in enumerate(bounds):
        # (the conversions of the four blocks on four worker threads side by side: 3.9-4.1 ms for the stage instead of 3.2 --
        # thread wake-ups and the GIL cost more than the 0.25 ms conversions they overlap; measured, dropped)
        blk = pmvo.stage_upload("filter%d" % k, points[lo:hi])
        # (the rows of a block are taken cell by cell -- 64 spatial neighbours per wave -- not in the candidates' raster order)
        order = spatial_order(blk)
        _lib.check(L.mh_filter_points_ordered(ctx, _lib.ptr(blk), hi - lo, pmvo._side, float(pmvo.conf_threshold),
                                              float(pmvo.visible_threshold), off(surf, lo), off(filt, lo), None, None,
                                              num_sub_p, lo, total, _lib.ptr(order), st), "mh_filter_points_ordered")
        _lib.check(L.mh_select_rows(ctx, off(surf, lo), None, 0, hi - lo, _lib.ptr(blk), None, _lib.ptr(out), None, None,
                                    off(cnt, k), off(cnt, k + 1), _lib.ptr(scratch), scratch.numel(), st), "mh_select_rows")
    flags = torch.empty((2, total), dtype=torch.uint8, pin_memory=True)
    hcnt = torch.empty((1,), dtype=torch.int32, pin_memory=True)
    flags[0].copy_(surf, non_blocking=True)
    flags[1].copy_(filt, non_blocking=True)
    hcnt.copy_(cnt[len(bounds):], non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    S = int(hcnt[0])
    pts = torch.empty((S, 3), dtype=torch.float32, pin_memory=True)
    pts.copy_(out[:S], non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()
    return flags.numpy(), pts.numpy()


def filter_negative_points(points, pmvo, args, step=30):
    """PMVO.py:535-557: surface / shell classification of the raw candidates, in N//30-sized pieces."""
    from . import dist as mdist

    if points.shape[0] % step != 0:
        step = step + 1
    num_sub_p = points.shape[0] // 30
    if mdist.world() == 1 and num_sub_p > 0 and isinstance(points, np.ndarray) and os.environ.get("MH_FILTER_BLOCKS", "1") != "0":
        flags, surface_points = _filter_single(points, pmvo, step, num_sub_p)
        surface_indexs, filter_indexs = flags[0].astype(bool), flags[1].astype(bool)
        print("surface_num:", surface_points.shape[:])
        print("num filter_unvisible:", np.sum(filter_indexs))
        return surface_indexs, surface_points, filter_indexs
    pieces = [points[i * num_sub_p:(i + 1) * num_sub_p] for i in range(step)]

    def work(sub):
        # the two vote arrays as the kernel writes them (uint8); the numpy piece is cast to float32 on the host like every
        # other entry point (PMVO.py:40), the gathered surface points of filter_points are not needed here
        _, (s, f, _, _) = pmvo._votes(sub, (True, True, False, False), raw=True)
        return torch.stack([s, f], 1)

    flags = mdist.map_chunks(pieces, work, pmvo.device, empty=lambda: torch.empty((0, 2), dtype=torch.uint8,
                                                                               device=pmvo.device))
    flags = torch.cat(flags, 0).cpu().numpy().astype(bool)
    surface_indexs, filter_indexs = flags[:, 0], flags[:, 1]
    covered = points[:flags.shape[0]]
    surface_points = covered[surface_indexs].astype(np.float32)   # the reference returns the fp32 copies
    print("surface_num:", surface_points.shape[:])
    print("num filter_unvisible:", np.sum(filter_indexs))
    return surface_indexs, surface_points, filter_indexs


def _start_mat_prefault(pmvo, args, points):
    """Ori3D.mat / Occ3D.mat of refine() are created and their pages made resident by a background thread (SparseMatWriter:
    26 ms of page faults at the headline size).  With refine's device-resident pass the whole of refine takes less than
    that, so the thread starts HERE, while optimize()'s iterations run: every surface point is a candidate voxel.  refine()
    adopts the writer when it is asked for the same directory and the default grid; a writer that is not adopted removes
    its temporary files when it is dropped."""
    from . import dist as mdist
    from . import pmvo_utils as U

    old = getattr(pmvo, "_mat_early", None)
    pmvo._mat_early = None
    if old is not None:
        old[1].abort()
    path = getattr(args, "save_path", "") or ""
    if mdist.rank() != 0 or not os.path.isdir(path) or os.environ.get("MH_MAT_EARLY", "1") == "0" or not len(points):
        return
    pmvo._mat_early = (path, U.SparseMatWriter(path, U.GRID_RESOLUTION, points, U.VOXEL_MIN, U.VOXEL_SIZE))


def _optimize_single(pts_np, pmvo, args):
    """optimize() on one rank.  The chunks rotate over three HIP streams (consecutive chunks are independent: the tail of
    one chunk's search kernel -- workgroups of points that see many views -- overlaps the front end and the head of the
    next chunks); every chunk's search writes straight into its slice of three device buffers.  The result files are
    STREAMED: a copy stream brings the rows of each group of eight finished chunks to pinned host arrays while later
    chunks compute, and the host appends them to optimize/*.npy (np.save's bytes, NpyRowWriter) -- when the last search
    ends, one group is left to copy and write.  While the GPU iterates, what refine() needs of the points alone is
    prepared on a low-priority stream (PMVO.start_refine_prefetch)."""
    from concurrent.futures import ThreadPoolExecutor

    from .pmvo_utils import NpyRowWriter

    num_sub_p = 5000
    M = int(pts_np.shape[0])
    step = M // num_sub_p + 1
    dev = pmvo.device
    streams = pmvo.side_streams(3)     # kept on the object: their tap-list scratch (1.2 GB each) is reused
    main = torch.cuda.current_stream(dev)
    cp = pmvo.aux_stream("copy")
    o_all = torch.empty((M, 3), dtype=torch.float32, device=dev)
    l_all = torch.empty((M,), dtype=torch.float32, device=dev)
    h_all = torch.empty((M,), dtype=torch.bool, device=dev)
    # the arrays the caller gets: pinned, filled group by group (torch's host allocator recycles the blocks)
    ho = torch.empty((M, 3), dtype=torch.float32, pin_memory=True)
    hl = torch.empty((M,), dtype=torch.float32, pin_memory=True)
    hh = torch.empty((M,), dtype=torch.bool, pin_memory=True)
    done = {}                          # chunk -> event on its stream

    def launch(dev_all, lo_chunk, hi_chunk):
        for st in streams:
            st.wait_stream(main)      # the copy of these rows (and whatever produced the maps) has been enqueued on `main`
        for i in range(lo_chunk, hi_chunk):
            a, b = i * num_sub_p, min((i + 1) * num_sub_p, M)
            if b <= a:
                continue
            st = streams[i % len(streams)]
            with torch.cuda.stream(st):
                pmvo.forward(dev_all[a:b], out=(o_all[a:b], l_all[a:b], h_all[a:b]))
                done[i] = torch.cuda.Event()
                done[i].record(st)

    nhead = len(streams)               # chunks launched before the rest of the points is converted and copied
    # the candidate points go to the device ONCE (3.4 MB at the headline size); a chunk is a slice of that tensor
    dev_all, staged = pmvo.upload_all_points(pts_np, head=nhead * num_sub_p, after_head=lambda d: launch(d, 0, nhead))
    # (the prefetch's helper thread starts BEFORE the bulk of the forwards is queued: its first launches then sit early in the
    # hardware queues and its host-side synchronisations pass while the GPU iterates -- 52.0-52.4 ms per pass against 52.4-53.6
    # when it was started behind them)
    pmvo.start_refine_prefetch(dev_all, num_sub_p, 100)
    launch(dev_all, nhead, step)
    _start_mat_prefault(pmvo, args, pts_np)
    # select_p.npy is the float32 copy of the input (PMVO.py:40,575): written while the GPU works
    select_points = pts_np if (pts_np.dtype == np.float32 and pts_np.flags.c_contiguous) else staged.copy()
    os.makedirs(args.save_root, exist_ok=True)
    pool = ThreadPoolExecutor(1)
    early = pool.submit(np.save, args.save_root + "/select_p.npy", select_points)
    # groups of finished chunks -> pinned host rows -> files
    files = (NpyRowWriter(args.save_root + "/select_o.npy", (M, 3), np.float32),
             NpyRowWriter(args.save_root + "/min_loss.npy", (M,), np.float32),
             NpyRowWriter(args.save_root + "/high_conf_index.npy", (M,), np.bool_))
    chunks = sorted(done)
    GROUP = 8
    groups = []
    with torch.cuda.stream(cp):
        for g0 in range(0, len(chunks), GROUP):
            grp = chunks[g0:g0 + GROUP]
            for i in grp[-len(streams):]:                 # the last chunk of the group on each stream
                cp.wait_event(done[i])
            a, b = grp[0] * num_sub_p, min((grp[-1] + 1) * num_sub_p, M)
            ho[a:b].copy_(o_all[a:b], non_blocking=True)
            hl[a:b].copy_(l_all[a:b], non_blocking=True)
            hh[a:b].copy_(h_all[a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cp)
            groups.append((a, b, ev))
    select_ori, min_loss, high_conf_index = ho.numpy(), hl.numpy(), hh.numpy()
    for a, b, ev in groups:
        ev.synchronize()
        files[0].write(select_ori[a:b])
        files[1].write(min_loss[a:b])
        files[2].write(high_conf_index[a:b])
    for f in files:
        f.close()
    for st in streams:                 # (all their work is over: the last group's copy waited for it)
        main.wait_stream(st)
    main.wait_stream(cp)
    early.result()
    pool.shutdown()
    return select_points, select_ori, min_loss, high_conf_index


def optimize(points, pmvo, args):
    """PMVO.py:565-595: forward() over chunks of 5000 points, results to optimize/*.npy.
    Returns (select_points, select_ori, min_loss, high_conf_index) -- the arrays the reference loads back from the files
    (PMVO.py:868-870); on one rank they live in pinned host memory."""
    from . import dist as mdist

    num_sub_p = 5000
    step = points.shape[0] // num_sub_p + 1
    pts_np = points if isinstance(points, np.ndarray) else torch.as_tensor(points).detach().cpu().numpy()
    if mdist.world() == 1:
        out = _optimize_single(pts_np, pmvo, args)
        assert len(out[2]) == len(out[0])
        return out
    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(4)
    streams = pmvo.side_streams(3)
    counter = [0]
    main = torch.cuda.current_stream()

    def join():                        # results are read on the main stream: join the side streams first
        for st in streams:
            main.wait_stream(st)

    # ONE float64 -> float32 conversion (numpy's rounding = the reference's `.type(torch.float)`, PMVO.py:40), in the pinned
    # staging buffer: the result is both what is uploaded and select_p.npy (PMVO.py:575)
    dev_all, select_points = pmvo.upload_all_points(pts_np)
    select_points = select_points.copy()
    early = None
    if mdist.rank() == 0:
        os.makedirs(args.save_root, exist_ok=True)
        early = pool.submit(np.save, args.save_root + "/select_p.npy", select_points)
    for st in streams:
        st.wait_stream(main)
    chunks = [dev_all[i * num_sub_p:(i + 1) * num_sub_p] for i in range(step)]

    def work(sub):
        st = streams[counter[0] % len(streams)]
        counter[0] += 1
        with torch.cuda.stream(st):
            _, o, l, h = pmvo.forward(sub)
            # (the points forward() returns are the float32 copy of its input: they do not travel back)
            out = torch.cat([o, l[:, None], h[:, None].to(torch.float32)], 1)
        out.record_stream(main)
        return out

    res = mdist.map_chunks(chunks, work, pmvo.device,
                           empty=lambda: torch.empty((0, 5), dtype=torch.float32, device=pmvo.device), after=join)
    res = torch.cat(res, 0).cpu().numpy()
    select_ori, min_loss = res[:, 0:3], res[:, 3]
    high_conf_index = res[:, 4] > 0.5
    assert len(min_loss) == len(select_points)
    if mdist.rank() == 0:
        jobs = (("select_o", select_ori), ("min_loss", min_loss), ("high_conf_index", high_conf_index))
        # the files are written side by side (numpy releases the GIL in write)
        list(pool.map(lambda j: np.save(args.save_root + "/%s.npy" % j[0], np.ascontiguousarray(j[1])), jobs))
        early.result()
    pool.shutdown()
    mdist.barrier()
    return select_points, select_ori, min_loss, high_conf_index


def _knn(data_points, query_points, k, device, mode="device", int32=False, self_query=False, keep_grid=None):
    """The `KDTree(data).query(queries, k)` of refine (PMVO.py:605,612,660,671) -> index [Q,k] int64 tensor on `device`.
    mode "device": exact grid k-NN kernel (csrc/knn.hip, scipy's result and order); "host": scipy on all cores.
    scipy returns index n for missing neighbours when k > n (the reference would raise on ori[index]); k is clamped.
    keep_grid: a list that receives the GridKNN object (device mode), so that a later query of a subset can reuse it."""
    k = min(k, data_points.shape[0])
    if mode == "device":
        from .pmvo_utils import GridKNN

        g = GridKNN(data_points, k_hint=k, device=device)
        if keep_grid is not None:
            keep_grid.append(g)
        return g.query(query_points, k, int32=int32, self_query=self_query)
    from scipy.spatial import KDTree

    _, index = KDTree(data=data_points).query(query_points, k, workers=-1)
    index = np.asarray(index).reshape(len(query_points), k)
    return torch.from_numpy(index.astype(np.int32) if int32 else index).to(device)


def _refine_device(points, ori, loss, pmvo, filter_unvisible_points, args, threshold, voxel_min, voxel_size,
                   grid_resolution):
    """The smoothing loop, the loss threshold, the shell points and the voxel fit of refine() (PMVO.py:602-726) on one rank
    with every array resident on the device from the first launch to the last: between the stages the host reads four
    counters, nothing else -- the `np.where` / `ori[index]` / `np.concatenate` steps of the reference are stable
    compactions on the device (mh_flag_less, mh_select_rows, mh_segment_heads), the neighbour table / head votes / scalp
    test of the points come from optimize()'s prefetch when it was made for these points, and the refine/*.npy files are
    written by a worker thread from pinned copies while the shell stage and the fit run.  Same kernels on the same values
    in the same order as the host-driven form below (tests pin both to the reference's multi-chunk run).

    -> dict(ori=..., loss=... host arrays of the smoothed result (also stored into the caller's arrays), grid, saver,
    save_error, and -- unless the shell stage has to take the host path (fewer kept points than neighbours asked for, or
    a query the grid could not finish at its first cell size) -- vox, vori, shell_points, shell_ori)."""
    import threading

    from . import pmvo_utils as U

    device = pmvo.device
    L, ctx = pmvo._L, pmvo._ctx
    n_all, sub_num = int(points.shape[0]), 5000
    step = n_all // sub_num + 1
    main = torch.cuda.current_stream(device)
    side = pmvo.aux_stream("refine_side")
    cp = pmvo.aux_stream("copy")
    off = lambda t, row, width=1: ctypes.c_void_p(t.data_ptr() + row * width * t.element_size())   # noqa: E731
    fu = np.ascontiguousarray(filter_unvisible_points) if filter_unvisible_points is not None else np.zeros((0, 3), np.float32)
    F = int(len(fu))
    T_up = stage("refine: uploads + prefetch adoption", device).__enter__()
    pts_dev = pmvo.stage_upload("refine_p", np.asarray(points).reshape(-1, 3))
    ori_dev = pmvo.stage_upload("refine_o", np.asarray(ori).reshape(-1, 3))
    loss_dev = pmvo.stage_upload("refine_l", np.asarray(loss).reshape(-1))
    pf = pmvo.take_refine_prefetch(pts_dev, sub_num, 100)
    pmvo.last_refine = {"device_pass": True, "prefetch_adopted": pf is not None, "shell_stage": "device"}   # (what ran: tests, bench)
    T_up.__exit__()
    # the shell points' votes and scalp test need the shell points only: first thing on the side stream
    hd = ht = fb_dev = fq_dev = None
    side.wait_stream(main)
    if F:
        with torch.cuda.stream(side):
            fb_dev = pmvo.stage_upload("refine_fb", fu.reshape(-1, 3))                    # float32 (votes, scalp test, files)
            fq_dev = pmvo.stage_upload("refine_fq", fu.reshape(-1, 3), np.float64) if fu.dtype == np.float64 else fb_dev
            hd = torch.empty((F,), dtype=torch.uint8, device=device)
            _lib.check(L.mh_filter_points(ctx, _lib.ptr(fb_dev), F, pmvo._side, float(pmvo.conf_threshold),
                                          float(args.PMVO.visible_threshold), None, None, None, _lib.ptr(hd), 5000, 0, F,
                                          _lib.stream_ptr()), "mh_filter_points")
            ht = pmvo.head_top_mask_device(fb_dev)
    if pf is not None:
        grid, index_all, head_all, head_top_all = pf.grid, pf.index_all, pf.head_all, pf.head_top_all
    else:
        with stage("refine: knn (surface)", device):
            grid = U.GridKNN(pts_dev, k_hint=100, device=device)
            index_all = grid.query(pts_dev, min(100, n_all), int32=True, self_query=True).contiguous()
        with stage("refine: head-top mask", device):
            head_top_all = pmvo.head_top_mask_device(pts_dev)
        head_all = torch.empty((n_all,), dtype=torch.uint8, device=device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            _lib.check(L.mh_filter_points_ordered(ctx, _lib.ptr(pts_dev), n_all, pmvo._side, float(pmvo.conf_threshold),
                                                  float(pmvo.visible_threshold), None, None, None, _lib.ptr(head_all),
                                                  sub_num, 0, n_all, _lib.ptr(grid.cell_order()), _lib.stream_ptr()),
                       "mh_filter_points_ordered")
    T_loop = stage("refine: smoothing loop", device).__enter__()
    K = int(index_all.shape[1])
    st = _lib.stream_ptr()
    centers = torch.empty((n_all, 3), dtype=torch.float32, device=device)
    loss_all = torch.empty((n_all,), dtype=torch.float32, device=device)
    # Only the ORIENTATIONS chain from chunk to chunk (chunk k+1's medoids read what chunk k replaced, PMVO.py:614,640):
    # medoid -> replacement rule, two small launches per chunk on the main stream; the loss of a chunk's medoid directions
    # (PMVO.py:619-623) feeds nothing in later chunks and runs beside the chain on the side stream, in groups of chunks as
    # soon as their medoids exist (short groups first, so that the side stream starts early); one launch writes every loss.
    g0, nxt, grp = 0, 2, 2
    with torch.cuda.stream(side):
        st2 = _lib.stream_ptr()
    # (the chain on a high-priority stream of its own: measured 4.1-4.4 ms against 3.6 on the caller's stream, docs/HISTORY.md)
    chain = main
    chain.wait_stream(main)
    stc = ctypes.c_void_p(chain.cuda_stream)
    for i in range(step):
        lo, hi = i * sub_num, min((i + 1) * sub_num, n_all)
        if hi > lo:
            _lib.check(L.mh_medoid_indexed(ctx, _lib.ptr(ori_dev), off(index_all, lo, K), hi - lo, K, off(centers, lo, 3),
                                           None, stc), "mh_medoid_indexed")
            _lib.check(L.mh_replace_dissimilar(ctx, off(centers, lo, 3), off(ori_dev, lo, 3), 0.95, hi - lo, stc),
                       "mh_replace_dissimilar")
        if (i + 1 == nxt or i == step - 1) and hi > g0:
            ev = torch.cuda.Event()
            ev.record(chain)
            side.wait_event(ev)
            _lib.check(L.mh_refine_loss_maps(ctx, off(pts_dev, g0, 3), off(centers, g0, 3), 0.005, 4.0, hi - g0, pmvo._side,
                                             float(pmvo.conf_threshold), off(loss_all, g0), None, sub_num, g0, n_all, st2),
                       "mh_refine_loss_maps")
            g0 = hi
            grp = min(grp * 2, 8)
            nxt = i + 1 + grp
    main.wait_stream(chain)
    main.wait_stream(side)
    _lib.check(L.mh_refine_combine(ctx, _lib.ptr(centers), _lib.ptr(loss_all), _lib.ptr(head_all), _lib.ptr(head_top_all),
                                   0.95, None, _lib.ptr(loss_dev), n_all, st), "mh_refine_combine")
    # the smoothed arrays travel to pinned host memory on the copy stream while the shell stage runs; a worker writes the
    # three files of PMVO.py:645-648 from them (the reference reads them back right away, :650-652 -- the arrays in memory
    # are what np.load would return)
    ev_loop = torch.cuda.Event()
    ev_loop.record(main)
    ho = torch.empty((n_all, 3), dtype=torch.float32, pin_memory=True)
    hl = torch.empty((n_all,), dtype=torch.float32, pin_memory=True)
    ev_res = torch.cuda.Event()
    with torch.cuda.stream(cp):
        cp.wait_event(ev_loop)
        ho.copy_(ori_dev, non_blocking=True)
        hl.copy_(loss_dev, non_blocking=True)
        ev_res.record(cp)
    new_ori, new_loss = ho.numpy(), hl.numpy()
    save_error = []
    from . import dist as mdist

    is_root = mdist.rank() == 0          # (several ranks, refine not sharded: all compute, rank 0 writes)
    if is_root:
        os.makedirs(args.output_path + "/refine", exist_ok=True)

    def _save():
        try:
            if is_root:
                np.save(args.output_path + "/refine/select_p.npy", points)
            ev_res.synchronize()
            if is_root:
                np.save(args.output_path + "/refine/select_o.npy", new_ori)
                np.save(args.output_path + "/refine/min_loss.npy", new_loss)
        except BaseException as e:      # re-raised on the calling thread after the join
            save_error.append(e)

    saver = threading.Thread(target=_save)
    saver.start()
    T_loop.__exit__()
    out = dict(grid=grid, saver=saver, save_error=save_error, ori=new_ori, loss=new_loss, ev_res=ev_res)
    # ---- loss threshold, shell points, concatenation: flags and compactions, no host in between (PMVO.py:651-693)
    T_shell = stage("refine: shell points", device).__enter__()
    valid = torch.empty((n_all,), dtype=torch.uint8, device=device)
    _lib.check(L.mh_flag_less(ctx, _lib.ptr(loss_dev), float(threshold), n_all, _lib.ptr(valid), st), "mh_flag_less")
    cap = n_all + F
    sel_p = torch.empty((cap, 3), dtype=torch.float32, device=device)
    sel_o = torch.empty((cap, 3), dtype=torch.float32, device=device)
    cnt = torch.zeros((2,), dtype=torch.int32, device=device)
    scratch = torch.empty(int(L.mh_select_scratch_bytes(max(n_all, F))), dtype=torch.uint8, device=device)
    _lib.check(L.mh_select_rows(ctx, _lib.ptr(valid), None, 0, n_all, _lib.ptr(pts_dev), _lib.ptr(ori_dev), _lib.ptr(sel_p),
                                _lib.ptr(sel_o), None, None, off(cnt, 0), _lib.ptr(scratch), scratch.numel(), st),
               "mh_select_rows")
    hcnt = torch.empty((2,), dtype=torch.int32, pin_memory=True)
    hcnt[:1].copy_(cnt[:1], non_blocking=True)
    main.synchronize()                   # (1) the number of points the threshold keeps
    n_valid = int(hcnt[0])
    n_sel = n_valid
    if F and n_valid:
        if n_valid < min(100, n_all):    # the reference then asks for n_valid neighbours: the host-driven stage below does
            pmvo.last_refine["shell_stage"] = "host (fewer kept points than neighbours)"
            T_shell.__exit__()
            return out
        # KDTree(select_points).query(fu, 100) on the grid of ALL points with the kept ones flagged valid: the indices
        # come back in terms of `points` (order-preserving compaction: same (distance, index) order), so the medoid reads
        # the full orientation array (PMVO.py:660-672)
        idx, status = grid.query_nosync(fq_dev, 100, valid_dev=valid)
        cen = torch.empty((F, 3), dtype=torch.float32, device=device)
        _lib.check(L.mh_medoid_indexed(ctx, _lib.ptr(ori_dev), _lib.ptr(idx), F, int(idx.shape[1]), _lib.ptr(cen), None, st),
                   "mh_medoid_indexed")
        main.wait_stream(side)
        # kept = not filter_head_points = not (head votes and not under the scalp top) (PMVO.py:674-686)
        _lib.check(L.mh_select_rows(ctx, _lib.ptr(hd), _lib.ptr(ht), 1, F, _lib.ptr(fb_dev), _lib.ptr(cen), _lib.ptr(sel_p),
                                    _lib.ptr(sel_o), None, off(cnt, 0), off(cnt, 1), _lib.ptr(scratch), scratch.numel(), st),
                   "mh_select_rows")
        hst = torch.empty((F,), dtype=torch.int32, pin_memory=True)
        hst.copy_(status, non_blocking=True)
        hcnt[1:].copy_(cnt[1:], non_blocking=True)
        main.synchronize()               # (2) kept shell rows; queries the grid could not finish at its first cell size
        if bool(hst.numpy().any()):
            # some shell points need another cell size (kept points sparser than the grid was laid out for): GridKNN's own
            # retries fix those rows of the table in place, then the medoids and the selection of the shell rows once more
            pmvo.last_refine["shell_stage"] = "device (%d queries retried on another cell size)" % int((hst.numpy() != 0).sum())
            grid.finish_nosync(fq_dev, 100, idx, hst.numpy(), valid_dev=valid)
            _lib.check(L.mh_medoid_indexed(ctx, _lib.ptr(ori_dev), _lib.ptr(idx), F, int(idx.shape[1]), _lib.ptr(cen), None, st),
                       "mh_medoid_indexed")
            _lib.check(L.mh_select_rows(ctx, _lib.ptr(hd), _lib.ptr(ht), 1, F, _lib.ptr(fb_dev), _lib.ptr(cen), _lib.ptr(sel_p),
                                        _lib.ptr(sel_o), None, off(cnt, 0), off(cnt, 1), _lib.ptr(scratch), scratch.numel(), st),
                       "mh_select_rows")
            hcnt[1:].copy_(cnt[1:], non_blocking=True)
            main.synchronize()
        n_sel = int(hcnt[1])
    else:
        main.wait_stream(side)
    T_shell.__exit__()
    # (no shell stage without kept points or without shell points, PMVO.py:658)
    hp = torch.empty((n_sel - n_valid, 3), dtype=torch.float32, pin_memory=True)
    hq = torch.empty((n_sel - n_valid, 3), dtype=torch.float32, pin_memory=True)
    ev_shell = torch.cuda.Event()
    with torch.cuda.stream(cp):
        cp.wait_stream(main)
        hp.copy_(sel_p[n_valid:n_sel], non_blocking=True)
        hq.copy_(sel_o[n_valid:n_sel], non_blocking=True)
        ev_shell.record(cp)
    with stage("refine: voxel fit + reduce", device):
        vox, vori = U.voxel_fit_device(sel_p, sel_o, n_sel, device, voxel_min, voxel_size, grid_resolution)
    ev_shell.synchronize()
    out.update(vox=vox, vori=vori, shell_points=hp.numpy(), shell_ori=hq.numpy())
    return out


def refine(points, ori, loss, pmvo, filter_unvisible_points, args, infer_inner=True, threshold=0.001,
           genrate_ori_only=False, voxel_min=None, voxel_size=None, grid_resolution=None, return_dense=True):
    """PMVO.py:602-764: KNN-medoid smoothing (sequentially dependent 5000-point chunks, in place), threshold,
    orientations for the occluded shell points, voxel fit, Ori3D.mat / Occ3D.mat.
    Returns the dense (occ [X,Y,Z], ori [X,Y,Z,3]) float64 arrays of the reference (it returns nothing itself);
    return_dense=False skips building them (PMVO.py's command line does: the files are written from the voxel list).
    voxel_min/voxel_size/grid_resolution default to the reference's hard-coded 256x256x192 @ 2.5 mm grid.
    args.knn = "host" switches the neighbour queries back to scipy's KDTree (default: the device kernel)."""
    from . import dist as mdist
    from . import pmvo_utils as U

    device = pmvo.device
    voxel_min = U.VOXEL_MIN if voxel_min is None else np.asarray(voxel_min, dtype=np.float64)
    voxel_size = U.VOXEL_SIZE if voxel_size is None else voxel_size
    grid_resolution = U.GRID_RESOLUTION if grid_resolution is None else np.asarray(grid_resolution).astype(np.int32)
    is_root = mdist.rank() == 0
    grid_all = []          # the spatial index of ALL points (device k-NN), reused for the shell query below
    # Ori3D.mat / Occ3D.mat are created NOW and their pages made resident in the background (every point that can end up in the
    # volume is known: the surface points and the shell candidates); the occupied elements are stored at the end
    mat_writer = None
    early, pmvo._mat_early = getattr(pmvo, "_mat_early", None), None
    if early is not None:           # optimize() started the files of this directory while its iterations ran
        if (is_root and early[0] == (getattr(args, "save_path", "") or "") and voxel_size == U.VOXEL_SIZE
                and np.array_equal(voxel_min, U.VOXEL_MIN) and np.array_equal(grid_resolution, U.GRID_RESOLUTION)):
            mat_writer = early[1]
        else:
            early[1].abort()
    if mat_writer is None and is_root and os.path.isdir(getattr(args, "save_path", "") or ""):
        cands = [np.asarray(p_)[:, :3] for p_ in (points, filter_unvisible_points) if p_ is not None and len(p_)]
        mat_writer = U.SparseMatWriter(args.save_path, grid_resolution, np.concatenate(cands, 0) if cands else None,
                                       voxel_min, voxel_size)
    dev_out = None
    pmvo.last_refine = {"device_pass": False, "prefetch_adopted": False, "shell_stage": "host"}
    if not genrate_ori_only:
        print("filter nosiy points...")
        # Device k-NN: the whole of :602-726 runs device-resident (_refine_device) -- on one rank, and on EVERY rank of a
        # multi-rank run that does not shard refine (the default: no communication, rank 0 writes the files).
        # MH_REFINE_DEVICE=0, MH_REFINE_SHARD=1, args.knn = "host" or MH_REFINE_CHAIN=0 take the host-driven forms below
        # (tests pin all of them to the reference).
        if ((mdist.world() == 1 or not mdist.refine_sharded()) and len(points) and getattr(args, "knn", "device") == "device"
                and os.environ.get("MH_REFINE_DEVICE", "1") != "0" and os.environ.get("MH_REFINE_CHAIN", "1") != "0"):
            dev_out = _refine_device(points, ori, loss, pmvo, filter_unvisible_points, args, threshold, voxel_min,
                                     voxel_size, grid_resolution)
            saver, save_error = dev_out["saver"], dev_out["save_error"]
            grid_all.append(dev_out["grid"])
            dev_out["ev_res"].synchronize()
            ori[:] = dev_out["ori"]              # in place, like the reference's chunk write-backs (PMVO.py:640-642)
            loss[:] = dev_out["loss"]
        if dev_out is None:
            # Neighbour indices and the head-top mask depend on the points only: ONE host query for all chunks (all
            # cores), then the sequentially dependent smoothing loop (chunk k+1 reads the orientations chunk k wrote,
            # PMVO.py:614,640) runs entirely on the device, stream-ordered, without a host round trip per chunk.
            n_all = points.shape[0]
            sub_num = 5000
            step = n_all // sub_num + 1
            # With several ranks (mdist.refine_sharded) every rank owns a fixed slice of EVERY chunk: slice k of a chunk of n
            # points is rows [lo + k*s, lo + (k+1)*s) with s = ceil(n / ranks).  A point's new orientation depends on the
            # orientations as they were when its chunk started (the medoid launch reads them before the chunk's write-back,
            # PMVO.py:614,640), so the slices of a chunk are independent; after the chunk one in-place all_gather per array
            # makes every rank's copy complete again -- the arrays a rank holds at the start of a chunk are the single-rank
            # ones, bit for bit.  Neighbour queries are needed for the owned rows only.
            R, rk = (mdist.world(), mdist.rank()) if mdist.refine_sharded() else (1, 0)
            # MH_REFINE_CHAIN=0: one rank runs the four-launches-per-chunk form of the sharded path (tests pin BOTH forms to the
            # reference's multi-chunk run, tests/test_multichunk_gpu.py)
            chain = R == 1 and os.environ.get("MH_REFINE_CHAIN", "1") != "0"

            def own(i):
                lo, hi = i * sub_num, min((i + 1) * sub_num, n_all)
                s_ = -(-(hi - lo) // R) if hi > lo else 0
                return lo, hi, s_, min(lo + rk * s_, hi), min(lo + (rk + 1) * s_, hi)

            with stage("refine: knn (surface)", device):
                if R == 1:
                    index_all = _knn(points, points, 100, device, getattr(args, "knn", "device"), int32=True,
                                     self_query=True, keep_grid=grid_all).contiguous()
                    row_of = [i * sub_num for i in range(step)]
                else:
                    mine = [np.arange(own(i)[3], own(i)[4]) for i in range(step)]
                    row_of = np.concatenate([[0], np.cumsum([len(m) for m in mine])]).tolist()
                    qidx = np.concatenate(mine) if mine else np.zeros(0, np.int64)
                    index_all = _knn(points, points[qidx], 100, device, getattr(args, "knn", "device"), int32=True,
                                     keep_grid=grid_all).contiguous()
            T_loop = stage("refine: smoothing loop", device).__enter__()
            pts_dev = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device)
            with stage("refine: head-top mask", device):      # scalp half of filter_head_points, once for all chunks
                head_top_all = pmvo.head_top_mask_device(pts_dev)
            slack = R * (-(-sub_num // R)) if R > 1 else 0     # rows the in-place exchange may touch past the last chunk
            if slack:
                ori_dev = torch.zeros((n_all + slack, 3), dtype=torch.float32, device=device)
                loss_dev = torch.zeros((n_all + slack,), dtype=torch.float32, device=device)
                ori_dev[:n_all] = torch.from_numpy(ori).to(device).type(torch.float)
                loss_dev[:n_all] = torch.from_numpy(loss).to(device).type(torch.float)
            else:       # (one rank: no tensor operation beyond the upload -- first uses of torch kernels cost a one-shot run ~40 ms)
                ori_dev = torch.from_numpy(ori).to(device).type(torch.float).contiguous()
                loss_dev = torch.from_numpy(loss).to(device).type(torch.float).contiguous()
            # per chunk four launches and no tensor op: medoid over the neighbour rows, the loss of that direction straight
            # from the maps, the head-filter votes, and the tail (-1 / replacement / 0.5) in place
            K = index_all.shape[1]
            center = torch.empty((sub_num, 3), dtype=torch.float32, device=device)
            loss_u = torch.empty((sub_num,), dtype=torch.float32, device=device)
            head = torch.empty((sub_num,), dtype=torch.uint8, device=device)
            L, ctx, st = pmvo._L, pmvo._ctx, _lib.stream_ptr()
            off = lambda t, row, width=1: ctypes.c_void_p(t.data_ptr() + row * width * t.element_size())   # noqa: E731
            if chain:
                # One rank.  Only the ORIENTATIONS chain from chunk to chunk (chunk k+1's medoids read what chunk k replaced,
                # PMVO.py:614,640): medoid -> replacement rule, 2 small launches per chunk on the main stream.  The loss of a
                # chunk's medoid directions (PMVO.py:619-623) and the head-filter votes feed nothing in later chunks, so they
                # run beside the chain on a second stream -- the votes of all points in one launch, the losses in groups of
                # eight chunks as soon as their medoids exist -- and one launch writes every loss at the end.  Same kernels on
                # the same values as the four-launches-per-chunk form the sharded path keeps (tests compare the two bit for bit).
                main = torch.cuda.current_stream(device)
                side = pmvo.side_streams(1)[0]
                centers = torch.empty((n_all, 3), dtype=torch.float32, device=device)
                loss_all = torch.empty((n_all,), dtype=torch.float32, device=device)
                head_all = torch.empty((n_all,), dtype=torch.uint8, device=device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    st2 = _lib.stream_ptr()
                    # (batch arguments: the reference votes and sums per 5000-point chunk, PMVO.py:604-621 -- the kernels place
                    # every point in its chunk, see include/mh_pmvo.h: mh_refine_loss_maps)
                    _lib.check(L.mh_filter_points(ctx, _lib.ptr(pts_dev), n_all, pmvo._side, float(pmvo.conf_threshold),
                                                  float(pmvo.visible_threshold), None, None, None, _lib.ptr(head_all),
                                                  sub_num, 0, n_all, st2),
                               "mh_filter_points")
                GROUP, g0 = 8, 0
                for i in range(step):
                    lo, hi = i * sub_num, min((i + 1) * sub_num, n_all)
                    if hi > lo:
                        _lib.check(L.mh_medoid_indexed(ctx, _lib.ptr(ori_dev), off(index_all, lo, K), hi - lo, K,
                                                       off(centers, lo, 3), None, st), "mh_medoid_indexed")
                        _lib.check(L.mh_replace_dissimilar(ctx, off(centers, lo, 3), off(ori_dev, lo, 3), 0.95, hi - lo, st),
                                   "mh_replace_dissimilar")
                    if ((i + 1) % GROUP == 0 or i == step - 1) and hi > g0:
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                        _lib.check(L.mh_refine_loss_maps(ctx, off(pts_dev, g0, 3), off(centers, g0, 3), 0.005, 4.0, hi - g0,
                                                         pmvo._side, float(pmvo.conf_threshold), off(loss_all, g0), None,
                                                         sub_num, g0, n_all, st2),
                                   "mh_refine_loss_maps")
                        g0 = hi
                main.wait_stream(side)
                _lib.check(L.mh_refine_combine(ctx, _lib.ptr(centers), _lib.ptr(loss_all), _lib.ptr(head_all),
                                               _lib.ptr(head_top_all), 0.95, None, _lib.ptr(loss_dev), n_all, st),
                           "mh_refine_combine")
            for i in range(0 if chain else step):
                lo, hi, s_, a, b = own(i)
                if hi <= lo:
                    continue
                n = b - a
                if n > 0:
                    _lib.check(L.mh_medoid_indexed(ctx, _lib.ptr(ori_dev), off(index_all, row_of[i], K), n, K,
                                                   _lib.ptr(center), None, st), "mh_medoid_indexed")
                    _lib.check(L.mh_refine_loss_maps(ctx, off(pts_dev, a, 3), _lib.ptr(center), 0.005, 4.0, n,
                                                     pmvo._side, float(pmvo.conf_threshold), _lib.ptr(loss_u), None,
                                                     sub_num, a, n_all, st),
                               "mh_refine_loss_maps")
                    _lib.check(L.mh_filter_points(ctx, off(pts_dev, a, 3), n, pmvo._side, float(pmvo.conf_threshold),
                                                  float(pmvo.visible_threshold), None, None, None, _lib.ptr(head),
                                                  sub_num, a, n_all, st),
                               "mh_filter_points")
                    _lib.check(L.mh_refine_combine(ctx, _lib.ptr(center), _lib.ptr(loss_u), _lib.ptr(head),
                                                   off(head_top_all, a), 0.95, off(ori_dev, a, 3), off(loss_dev, a), n, st),
                               "mh_refine_combine")
                if R > 1:
                    mdist.all_gather_rows_inplace(ori_dev, lo, s_)
                    mdist.all_gather_rows_inplace(loss_dev, lo, s_)
            ori_dev, loss_dev = ori_dev[:n_all], loss_dev[:n_all]
            ori[:] = ori_dev.cpu().numpy()
            loss[:] = loss_dev.cpu().numpy()
            T_loop.__exit__()
            saver = None
            if is_root:
                # the three files of PMVO.py:645-648 are written by a worker thread while the shell stage runs; the reference
                # reads them back right away (:650-652) -- the arrays in memory are what np.load would return
                import threading

                os.makedirs(args.output_path + "/refine", exist_ok=True)
                held = (points, ori, loss)          # not written again before the join below

                save_error = []

                def _save():
                    try:
                        for name, arr in zip(("select_p", "select_o", "min_loss"), held):
                            np.save(args.output_path + "/refine/%s.npy" % name, arr)
                    except BaseException as e:      # re-raised on the calling thread after the join
                        save_error.append(e)

                saver = threading.Thread(target=_save)
                saver.start()
        min_loss = loss
    else:
        saver = None
        points = np.load(args.output_path + "/refine/select_p.npy")
        ori = np.load(args.output_path + "/refine/select_o.npy")
        min_loss = np.load(args.output_path + "/refine/min_loss.npy")
    late = []               # worker threads that must be over before refine() returns
    if dev_out is not None and "vox" in dev_out:
        print("compute points orientation near the surface... ")
        vox, vori = dev_out["vox"], dev_out["vori"]
        if is_root:
            import threading

            def _save_shell():
                try:
                    np.save(args.output_path + "/refine/filter_unvisible.npy", dev_out["shell_points"])
                    np.save(args.output_path + "/refine/filter_unvisible_ori.npy", dev_out["shell_ori"])
                except BaseException as e:
                    save_error.append(e)

            late.append(threading.Thread(target=_save_shell))
            late[-1].start()
        late.append(saver)
        saver = None
    else:
        index = np.where(min_loss < threshold)[0]
        select_ori = ori[index]
        select_points = points[index]

        # orientation of the occluded shell points from their 100 nearest kept neighbours (PMVO.py:662-686)
        print("compute points orientation near the surface... ")
        T_shell = stage("refine: shell points", device).__enter__()
        filter_unvisible_ori = np.zeros((0, 3), np.float32)
        select_filter_unvisible_points = np.zeros((0, 3), np.float32)
        if len(select_points) and len(filter_unvisible_points):
            fu = np.ascontiguousarray(filter_unvisible_points)
            use_grid = bool(grid_all) and grid_all[0].M == len(points)
            if use_grid:
                # KDTree(select_points).query(fu) on the grid that already exists for all points: the points kept by the loss
                # threshold are flagged valid, the indices come back in terms of `points` (order-preserving compaction: same
                # (distance, index) order), so the medoid reads the full orientation array
                valid = np.zeros(len(points), np.uint8)
                valid[index] = 1
                ori_rows = ori
            else:
                ori_rows = select_ori
            sel_ori_dev = torch.from_numpy(np.ascontiguousarray(ori_rows, dtype=np.float32)).to(device)

            def shell_block(fb, row0=0):
                """rows row0.. of `fu` -> device tensors (medoid orientation of the 100 nearest kept points [n,3], head-filter votes [n],
                head-top mask [n]).  The points are independent: one medoid launch and one vote launch for all of them (the
                reference's 5000-point chunks bound its memory -- and place a point in a batch of its sums over views, which the
                vote kernel is told: PMVO.py:662-672)."""
                with stage("refine: knn (shell)", device):
                    if use_grid:
                        idx = grid_all[0].query(fb, 100, int32=True, valid=valid).contiguous()
                    else:
                        idx = _knn(select_points, fb, 100, device, getattr(args, "knn", "device"), int32=True).contiguous()
                fb_dev = torch.from_numpy(fb.astype(np.float32)).to(device).contiguous()
                F, K = idx.shape
                cen = torch.empty((F, 3), dtype=torch.float32, device=device)
                hd = torch.empty((F,), dtype=torch.uint8, device=device)
                _lib.check(pmvo._L.mh_medoid_indexed(pmvo._ctx, _lib.ptr(sel_ori_dev), _lib.ptr(idx), F, K, _lib.ptr(cen), None,
                                                     _lib.stream_ptr()), "mh_medoid_indexed")
                _lib.check(pmvo._L.mh_filter_points(pmvo._ctx, _lib.ptr(fb_dev), F, pmvo._side, float(pmvo.conf_threshold),
                                                    float(args.PMVO.visible_threshold), None, None, None, _lib.ptr(hd),
                                                    5000, row0, len(fu), _lib.stream_ptr()), "mh_filter_points")
                ht = pmvo.head_top_mask_device(fb_dev)
                return cen, hd, ht

            if mdist.refine_sharded():           # block k of the shell points belongs to rank k; one all_gather of the results
                W_ = mdist.world()
                cuts = [(len(fu) * k) // W_ for k in range(W_ + 1)]

                def packed(fb, k):
                    cen, hd, ht = shell_block(fb, cuts[k])
                    out = torch.empty((cen.shape[0], 4), dtype=torch.float32, device=device)
                    out[:, :3] = cen
                    out[:, 3] = (~(hd.bool() & ~ht.bool())).to(torch.float32)
                    return out

                res = torch.cat(mdist.map_chunks([fu[cuts[k]:cuts[k + 1]] for k in range(W_)], packed, device,
                                                 empty=lambda: torch.empty((0, 4), dtype=torch.float32, device=device),
                                                 with_index=True), 0)
                res = res.cpu().numpy()
                keep = res[:, 3] > 0.5
                centres = res[:, :3]
            else:
                # one rank: the three results go to the host as they are (no tensor operation: in a one-shot process every
                # first use of a torch kernel costs tens of milliseconds of code-object loading)
                cen, hd, ht = shell_block(fu)
                keep = ~np.logical_and(hd.cpu().numpy().astype(bool), ~ht.cpu().numpy().astype(bool))     # not filter_head_points
                centres = cen.cpu().numpy()
            filter_unvisible_ori = np.ascontiguousarray(centres[keep])
            select_filter_unvisible_points = fu.astype(np.float32)[keep]
        T_shell.__exit__()
        if saver is not None:
            saver.join()
            if save_error:
                raise save_error[0]
        if is_root:
            np.save(args.output_path + "/refine/filter_unvisible.npy", select_filter_unvisible_points)
            np.save(args.output_path + "/refine/filter_unvisible_ori.npy", filter_unvisible_ori)

        select_ori = np.concatenate([select_ori, filter_unvisible_ori], 0)
        select_points = np.concatenate([select_points, select_filter_unvisible_points], 0)

        # voxel fit (PMVO.py:695-726): every rank fits a disjoint slab of voxels; one reduce assembles the volume
        # The volume stays a list of occupied voxels; the dense float64 arrays of the reference exist only on request.
        with stage("refine: voxel fit + reduce", device):
            if mdist.world() > 1 and not mdist.refine_sharded():
                # several ranks, refine not sharded: every rank holds every point -- each fits the whole volume itself, no
                # exchange (rank 0 writes); the slab exchange belongs to the sharded form
                res = U.voxel_fit(select_points, select_ori, device, voxel_min, voxel_size, grid_resolution, dense=False)
                vox, vori = res["voxels"].cpu().numpy(), res["ori"].cpu().numpy()
            else:
                vox, vori = mdist.voxel_fit_reduced(select_points, select_ori, device, voxel_min, voxel_size,
                                                    grid_resolution, sparse=True)

    if is_root:
        if infer_inner:
            # merge the network's orientation for points seen in <= 2 views (PMVO.py:733-751)
            coarse_data = np.load(args.data.root + "/ours/raw.npy")
            unvisible_index = pmvo.compute_unvisible_points(
                torch.from_numpy(coarse_data[:, :3].astype(np.float32)).to(device)).cpu().numpy()
            vox, vori, un_visible_points, unvisible_ori = U.merge_inner_points(vox, vori, coarse_data, unvisible_index,
                                                                               voxel_min, voxel_size, grid_resolution)
            np.save(os.path.join(args.save_path, "coarse.npy"), un_visible_points)
            np.save(os.path.join(args.save_path, "coarse_ori.npy"), unvisible_ori)
        with stage("refine: Ori3D/Occ3D.mat", device):
            if mat_writer is not None:
                mat_writer.finish(vox, vori)
            else:
                U.save_ori_occ_mat_sparse(args.save_path, grid_resolution, vox, vori)
    for t in late:
        t.join()
    if late and save_error:
        raise save_error[0]
    mdist.barrier()
    if not return_dense:
        return None
    return U.dense_from_sparse(grid_resolution, vox, vori)
