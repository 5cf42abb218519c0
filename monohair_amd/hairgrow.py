"""Strand tracing on the fitted orientation/occupancy volume -- host-side mirror of the tracing half of the
reference's `HairGrow.py::class HairGrowing` (__init__ :41-55, trace :59-149, traceFromScalp :154-223,
GenerateGuideStrandFromScalp :226-265, randomlyGenerateSegments :269-299, VoxelToWorld :816-824), the immediate
consumer of Ori3D.mat / Occ3D.mat.  Segment connection and scalp attachment
(HairGrow.py:303-786) are not part of this package.

All seeds are traced in parallel by the HIP kernels of csrc/hairgrow.hip; the sequential `flag` gate only decides
which finished traces are kept and is replayed afterwards (mh_strands_accept).  The jitter of every trace() call
comes from torch's CPU generator in the order the reference draws it, so a seeded run reproduces the reference's
CPU path bit for bit."""
import ctypes

import numpy as np
import torch

from . import _lib
from .pmvo_utils import _ctx_for, get_ground_truth_3D_occ, get_ground_truth_3D_ori, save_hair_strands, voxel_to_points


def _hp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class HairGrowing:
    def __init__(self, occ_path, ori_path, device="cuda:0", image_size=[1120, 1992], occ=None, ori=None):
        """occ_path/ori_path: Occ3D.mat / Ori3D.mat as PMVO writes them; or pass the readers' arrays directly
        (occ [Z,Y,X,1], ori [Z,Y,X,3])."""
        if not torch.cuda.is_available():
            raise _lib.MhError("HairGrowing needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device(device)
        self.image_size = image_size
        occ = get_ground_truth_3D_occ(occ_path) if occ is None else np.asarray(occ, np.float32)
        ori = get_ground_truth_3D_ori(ori_path) if ori is None else np.asarray(ori, np.float32)
        occ_t = torch.from_numpy(np.ascontiguousarray(occ[..., 0])).to(self.device).float()          # [Z,H,W]
        ori_t = torch.from_numpy(np.ascontiguousarray(ori)).to(self.device).float()                  # [Z,H,W,3]
        self.Z, self.H, self.W = occ_t.shape
        self._ctx = _ctx_for(self.device)
        self._vox = torch.empty((self.Z, self.H, self.W, 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_volume_pack(self._ctx, _lib.ptr(occ_t), _lib.ptr(ori_t), self.W, self.H, self.Z,
                                                 _lib.ptr(self._vox), _lib.stream_ptr()), "mh_volume_pack")
        # attribute surface of the reference (HairGrow.py:49-55)
        self.occ = self._vox[..., 3][None]                        # [1,Z,H,W]
        self.ori = self._vox[..., :3].permute(3, 0, 1, 2)         # [3,Z,H,W], y/z negated
        self.strands = None

    # ------------------------------------------------------------------ kernels
    def _trace_scalp(self, pts, nrm, thr):
        n = pts.shape[0]
        out = torch.empty((n, 257, 3), dtype=torch.float32, device=self.device)
        ln = torch.empty((n,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_trace_scalp(self._ctx, _lib.ptr(self._vox), self.W, self.H, self.Z, _lib.ptr(pts),
                                                 _lib.ptr(nrm), n, float(thr), _lib.ptr(out), _lib.ptr(ln),
                                                 _lib.stream_ptr()), "mh_trace_scalp")
        return out, ln

    def _trace_seeds(self, seeds, thr):
        n = seeds.shape[0]
        out = torch.empty((n, 513, 3), dtype=torch.float32, device=self.device)
        first = torch.empty((n,), dtype=torch.int32, device=self.device)
        ln = torch.empty((n,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_trace_seeds(self._ctx, _lib.ptr(self._vox), self.W, self.H, self.Z,
                                                 _lib.ptr(seeds), n, float(thr), _lib.ptr(out), _lib.ptr(first),
                                                 _lib.ptr(ln), _lib.stream_ptr()), "mh_trace_seeds")
        return out, first, ln

    def _accept(self, flag, pts, first, ln, seeds, mode):
        """sequential flag gate on the host over finished traces -> list of [L,3] numpy strands.  The fixed-stride rows
        (513 / 257 points per seed, mostly empty: 1.3 GB for 215 k seeds) are packed on the device first
        (mh_strands_compact), so only the points that exist cross PCIe."""
        n, stride = pts.shape[0], pts.shape[1]
        first_h = np.ascontiguousarray(first.cpu().numpy(), dtype=np.int32)
        ln_h = np.ascontiguousarray(ln.cpu().numpy(), dtype=np.int32)
        seeds_h = np.ascontiguousarray(seeds.cpu().numpy(), dtype=np.float32)
        lens = np.maximum(ln_h, 0).astype(np.int64)
        offs_h = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64) if n else np.zeros(0, np.int64)
        total = int(lens.sum())
        packed = torch.empty((max(total, 1), 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_strands_compact(self._ctx, _lib.ptr(pts), _lib.ptr(first.contiguous()),
                                                     _lib.ptr(ln.contiguous()), _lib.ptr(torch.from_numpy(offs_h).to(self.device)),
                                                     n, stride, _lib.ptr(packed), _lib.stream_ptr()), "mh_strands_compact")
        pts_h = np.ascontiguousarray(packed.cpu().numpy())
        assert total < 2 ** 31
        offs32 = offs_h.astype(np.int32)
        acc = np.zeros(n, np.uint8)
        _lib.check(_lib.lib().mh_strands_accept(self.W, self.H, self.Z, _hp(flag), _hp(pts_h), _hp(offs32), _hp(ln_h),
                                                0, _hp(seeds_h), n, mode, _hp(acc)), "mh_strands_accept")
        return [pts_h[offs_h[i]:offs_h[i] + ln_h[i]] for i in np.flatnonzero(acc)]

    def _voxel_rounds(self, flag, thrDot, rounds):
        """`rounds` passes of trace() over the occupied voxels.  The reference shifts its seed tensor IN PLACE on
        every call (HairGrow.py:62-63: += 0.5, += rand*0.5), so the shifts accumulate from round to round."""
        positive = torch.nonzero(self.occ[0], as_tuple=False)
        pos = torch.flip(positive, dims=[1]).type(torch.float)                        # (x,y,z)
        n = pos.shape[0]
        jitter = torch.rand(rounds * n, 3).to(self.device)       # CPU generator, the reference's draw order
        seeds = []
        for r in range(rounds):
            pos = pos + 0.5
            pos = pos + jitter[r * n:(r + 1) * n] * 0.5
            seeds.append(pos)
        seeds = torch.cat(seeds, 0).contiguous()
        out, first, ln = self._trace_seeds(seeds, thrDot)
        return self._accept(flag, out, first, ln, seeds, 0)

    def _to_device_views(self, strands_np):
        if not strands_np:
            return []
        cat = torch.from_numpy(np.concatenate(strands_np, 0)).to(self.device)
        return list(torch.split(cat, [s.shape[0] for s in strands_np]))

    # ------------------------------------------------------------------ reference methods
    def trace(self, seedPos, flag, thrDot, W=None, H=None, Z=None):
        """HairGrow.py:59-149 for ONE seed (the drivers above trace all seeds in one launch instead): shifts seedPos in
        place by 0.5 and by rand*0.5 like the reference, reads `flag` ([Z,H,W] tensor or array) at the seed only, and
        returns the strand [L,3] (L >= 5) or False."""
        seedPos += torch.tensor([0.5, 0.5, 0.5], dtype=torch.float, device=seedPos.device)
        seedPos += torch.rand_like(seedPos) * 0.5
        x, y, z = (min(max(int(v), 0), hi - 1) for v, hi in zip(seedPos.tolist(), (self.W, self.H, self.Z)))
        if float(flag[z, y, x]) >= 3:
            return False
        out, first, ln = self._trace_seeds(seedPos.to(self.device).type(torch.float)[None].contiguous(), thrDot)
        n, f = int(ln[0]), int(first[0])
        return out[0, f:f + n].clone() if n >= 5 else False

    def traceFromScalp(self, seedPos, seedNormal, thrDot, W=None, H=None, Z=None, pointsTree=None):
        """HairGrow.py:154-223 for ONE root: the strand [L,3], or None when it grows into the head."""
        out, ln = self._trace_scalp(seedPos.to(self.device).type(torch.float)[None].contiguous(),
                                    seedNormal.to(self.device).type(torch.float)[None].contiguous(), thrDot)
        n = int(ln[0])
        return out[0, :n].clone() if n > 0 else None

    def GenerateGuideStrandFromScalp(self, scalp_points, scalp_normals, pointsTree=None, thrDot=0.8):
        """HairGrow.py:226-265 -> (strands: list of [L,3] device tensors in voxel space, num_root)."""
        flag = np.zeros((self.Z, self.H, self.W), np.float32)
        sp = scalp_points.to(self.device).type(torch.float).contiguous()
        sn = scalp_normals.to(self.device).type(torch.float).contiguous()
        out, ln = self._trace_scalp(sp, sn, thrDot)
        roots = self._accept(flag, out, torch.zeros_like(ln), ln, sp, 1)
        num_root = len(roots)
        strands = roots + self._voxel_rounds(flag, thrDot, 2)
        self.strands = self._to_device_views(strands)
        return self.strands, num_root

    def randomlyGenerateSegments(self, thrDot=0.8):
        """HairGrow.py:269-299."""
        flag = np.zeros((self.Z, self.H, self.W), np.float32)
        self.strands = self._to_device_views(self._voxel_rounds(flag, thrDot, 3))
        return self.strands

    def VoxelToWorld(self, strands, bust_to_origin=None):
        """HairGrow.py:816-824."""
        out = []
        for ss in strands:
            w = voxel_to_points(ss.clone()).cpu().numpy()
            if bust_to_origin is not None:
                w -= bust_to_origin
            out.append(w)
        return out


def generate_segments(occ_path, ori_path, scalp_points_voxel, scalp_normals_voxel, save_path, bust_to_origin,
                      grow_threshold=0.8, device="cuda:0"):
    """The `generate_segments` stage of HairGrow.py's __main__ (:897-907, without the Laplacian smoothing pass):
    scalp_segment.hair + num_root.npy."""
    import os

    solver = HairGrowing(occ_path, ori_path, device=device)
    strands, num_root = solver.GenerateGuideStrandFromScalp(scalp_points_voxel, scalp_normals_voxel, None,
                                                            grow_threshold)
    world = solver.VoxelToWorld(strands, bust_to_origin)
    save_hair_strands(os.path.join(save_path, "scalp_segment.hair"), world, bust_to_origin, translate=False)
    np.save(os.path.join(save_path, "num_root.npy"), np.array(num_root))
    return world, num_root
