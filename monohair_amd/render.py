"""Renderers either side of the PMVO path, on the GPU (no OpenGL context): the depth-map producer (SURVEY.md §8f
rank 2) and the strand-segment renderer of infer_inner (rank 4, `render_data` at the end of this file).

Depth-map producer: the host mirror of the reference's
`Utils/Render_utils.py::render_bust_hair_depth` (:310-347) on top of `mh_render_depth` (csrc/raster.hip).

The reference draws the COLMAP hair mesh and the bust mesh with moderngl/EGL and the BustObj shader (:146-188) and
writes `render_depth/<view>.npy` = float32 [H,W,3], value (-z_cam / 2) * 255, background 255.  Here the same maps
come from a specified HIP rasteriser (no OpenGL context needed) and can stay on the device: `render_depth_planes`
returns a [V,H,W] tensor that `PMVO.from_planes` / `PMVO.from_u8` take as is.  The rasterisers follow the GL
specification and are pinned against a real OpenGL implementation (SwiftShader) up to what GL leaves to the driver
(sub-pixel snapping, interpolation rounding; docs/HISTORY.md §4.9, §4.10).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .camera import camera_records, load_cam, parsing_camera
from .pmvo_utils import _ctx_for, read_obj

BUST_TO_ORIGIN = np.array([0.006, -1.644, 0.010])      # Render_utils.py:312


class DepthRenderer:
    """Holds the meshes on the device (several meshes are concatenated in draw order, Render_utils.py:322-331)."""

    def __init__(self, meshes, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.MhError("monohair_amd.render needs a ROCm GPU: there is no CPU fallback")
        self.device = torch.device(device)
        verts, faces, base = [], [], 0
        for v, f in meshes:
            v = np.asarray(v, dtype=np.float64).reshape(-1, 3)
            f = np.asarray(f, dtype=np.int64).reshape(-1, 3)
            verts.append(v.astype(np.float32))
            faces.append((f + base).astype(np.int32))
            base += len(v)
        self.verts = torch.from_numpy(np.concatenate(verts) if verts else np.zeros((0, 3), np.float32)).to(self.device)
        self.faces = torch.from_numpy(np.concatenate(faces) if faces else np.zeros((0, 3), np.int32)).to(self.device)
        self._L = _lib.lib()
        self._ctx = _ctx_for(self.device)
        self._scratch = None

    def render(self, cam_record, H, W, pixel_center=0.5, out=None, channels=1):
        """One view -> float32 [H,W] (channels=1) or [H,W,channels] device tensor."""
        H, W = int(H), int(W)
        Nv, Nf = self.verts.shape[0], self.faces.shape[0]
        need = int(self._L.mh_render_scratch_bytes(Nv, Nf, H, W))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((H, W) if channels == 1 else (H, W, channels), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.numel() == H * W * channels and out.dtype == torch.float32
        rec = np.ascontiguousarray(cam_record, dtype=np.float32)
        with torch.cuda.device(self.device):
            _lib.check(self._L.mh_render_depth(self._ctx, rec.ctypes.data_as(ctypes.c_void_p), _lib.ptr(self.verts), Nv,
                                               _lib.ptr(self.faces), Nf, H, W, float(pixel_center),
                                               _lib.ptr(self._scratch), need, _lib.ptr(out), channels,
                                               _lib.stream_ptr()), "mh_render_depth")
        return out


def render_depth_planes(camera, meshes, image_size, device="cuda:0", pixel_center=0.5):
    """camera: dict view -> Camera; meshes: [(vertices, faces), ...] -> depth [V,H,W] float32 on `device`."""
    H, W = int(image_size[0]), int(image_size[1])
    r = DepthRenderer(meshes, device)
    recs = camera_records(camera)
    out = torch.empty((len(recs), H, W), dtype=torch.float32, device=r.device)
    for i in range(len(recs)):
        r.render(recs[i], H, W, pixel_center, out=out[i])
    return out


def render_bust_hair_depth(colmap_points_path, camera_path, save_root, image_size=[1280, 720], capture_imgs=False,
                           bust_path=None, Headless=True, device="cuda:0", pixel_center=0.5):
    """Same arguments and files as Render_utils.py:310-347: with capture_imgs, `<save_root>/<view>.npy`
    (float32 [H,W,3] = depth*255) and a `<view>.JPG` preview; otherwise `<save_root>/<view>/bust_hair_depth.png`."""
    from PIL import Image

    v, f = read_obj(colmap_points_path)
    meshes = [(v + BUST_TO_ORIGIN, f)]
    if bust_path is not None:
        bv, bf = read_obj(bust_path)
        meshes.append((bv + BUST_TO_ORIGIN, bf))
    camera = parsing_camera(load_cam(camera_path))
    H, W = int(image_size[0]), int(image_size[1])
    r = DepthRenderer(meshes, device)
    recs = camera_records(camera)
    for i, view in enumerate(camera.keys()):
        d3 = r.render(recs[i], H, W, pixel_center, channels=3).cpu().numpy()
        u8 = np.clip(np.rint(d3), 0, 255).astype(np.uint8)       # cv2.imwrite saturate-rounds floats to 8 bits
        if capture_imgs:
            os.makedirs(save_root, exist_ok=True)
            np.save(os.path.join(save_root, view + ".npy"), d3)
            Image.fromarray(u8).save(os.path.join(save_root, view + ".JPG"))
        else:
            os.makedirs(os.path.join(save_root, view), exist_ok=True)
            Image.fromarray(u8).save(os.path.join(save_root, view, "bust_hair_depth.png"))


# ------------------------------------------------------------------------------------------------------------------
# Strand-segment renderer: Utils/Render_utils.py:269-307 (render_data) -- the images infer_inner.py:60-73 renders from
# the segments traced on the exterior volume and hands to DeepMVSHair.
# ------------------------------------------------------------------------------------------------------------------
def strand_line_buffers(strands):
    """The two vertex buffers of the reference's StrandsObj (Render_utils.py:9-29): for every strand of n points the
    n-1 segments as consecutive vertex pairs (0,1),(1,2),... and, per vertex, the tangent strand[i+1]-strand[i] (the
    last point repeats the last difference) -> (Lines [2*S,3], tangent [2*S,3]) float32."""
    lines, tans = [], []
    for strand in strands:
        strand = np.asarray(strand, dtype=np.float64).reshape(-1, 3)
        n = strand.shape[0]
        if n < 2:
            continue
        tangent = np.concatenate([strand[1:] - strand[:-1], strand[-1:] - strand[-2:-1]], 0)
        index = np.stack([np.arange(0, n - 1), np.arange(1, n)], 1).reshape(-1)
        lines.append(strand[index])
        tans.append(tangent[index])
    if not lines:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
    return np.concatenate(lines).astype(np.float32), np.concatenate(tans).astype(np.float32)


class StrandRenderer:
    """Bust mesh + strand segments resident on the device; render(view, ...) = one draw + read-back of the reference's
    Renderer (Render_utils.py:206-266) with its clear colour and the two shader options."""

    LINE_WIDTH = 3          # StrandsObj.__init__: ctx.line_width = 3.0 (Render_utils.py:30)

    def __init__(self, strands, vertices, faces, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.MhError("monohair_amd.render needs a ROCm GPU: there is no CPU fallback")
        self.device = torch.device(device)
        lp, lt = strand_line_buffers(strands)
        self.nseg = len(lp) // 2
        self.line_pts = torch.from_numpy(lp).to(self.device)
        self.line_tan = torch.from_numpy(lt).to(self.device)
        self.verts = torch.from_numpy(np.asarray(vertices, dtype=np.float64).reshape(-1, 3).astype(np.float32)).to(self.device)
        self.faces = torch.from_numpy(np.asarray(faces, dtype=np.int64).reshape(-1, 3).astype(np.int32)).to(self.device)
        self._L = _lib.lib()
        self._ctx = _ctx_for(self.device)
        self._scratch = None

    def render(self, cam_record, H, W, color_option, depth_option, clear, draw_strands=True, pixel_center=0.5, out=None,
               line_width=None, line_rule=0):
        """-> float32 [H,W,3] device tensor in the shader's range (0..1).  line_width defaults to the reference's 3;
        line_rule 0 = OpenGL's diamond-exit rule, 1 = the end pixel of every segment too (the comparison with SwiftShader)."""
        H, W = int(H), int(W)
        Nv, Nf = self.verts.shape[0], self.faces.shape[0]
        need = int(self._L.mh_render_strands_scratch_bytes(Nv, Nf, self.nseg, H, W))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        rec = np.ascontiguousarray(cam_record, dtype=np.float32)
        width = self.LINE_WIDTH if line_width is None else int(line_width)
        with torch.cuda.device(self.device):
            _lib.check(self._L.mh_ctx_set_option(self._ctx, b"line_rule", int(line_rule)), "line_rule")
            try:
                _lib.check(self._L.mh_render_strands(
                    self._ctx, rec.ctypes.data_as(ctypes.c_void_p), _lib.ptr(self.verts), Nv, _lib.ptr(self.faces), Nf,
                    _lib.ptr(self.line_pts), _lib.ptr(self.line_tan), self.nseg, H, W, float(pixel_center), width,
                    int(color_option) if draw_strands else -1, int(depth_option), float(clear), _lib.ptr(self._scratch),
                    need, _lib.ptr(out), _lib.stream_ptr()), "mh_render_strands")
            finally:
                self._L.mh_ctx_set_option(self._ctx, b"line_rule", 0)
        return out


def _save_png(path, rgb01, swap):
    """cv2.imwrite(path, image * 255): float -> 8 bit with saturation and round-half-even; the reference passes the colour
    images as color[..., [2,1,0]] because OpenCV stores BGR, so the FILE holds the shader's (r, g, b)."""
    from PIL import Image

    u8 = np.clip(np.rint(rgb01 * 255.0), 0, 255).astype(np.uint8)
    del swap            # [2,1,0] followed by OpenCV's BGR->file order is the identity on the file's RGB
    Image.fromarray(u8).save(path)


def render_data(camera, strands, vertices, faces, image_size=[1280, 720], save_root=None, device="cuda:0",
                pixel_center=0.5):
    """Same arguments and files as Render_utils.py:269-307.  Per view, under <save_root>/<view>/:
      bust_depth.png          bust only, depth/2 on white                                   (:275-279)
      undirectional_map.png   strands coloured by their 2D direction (2 theta), bust black   (:283-289)
      mask.png                strands white, bust black                                      (:293-298)
      hair_depth.png          strands depth/2, bust white, on white                          (:300-306)
    (The depth images are grey, so OpenCV's channel order does not matter for them.)"""
    H, W = int(image_size[0]), int(image_size[1])
    r = StrandRenderer(strands, vertices, faces, device)
    recs = camera_records(camera)
    os.makedirs(save_root, exist_ok=True)
    passes = (("bust_depth.png", False, 0, 0, 1.0), ("undirectional_map.png", True, 2, 1, 0.0),
              ("mask.png", True, 3, 1, 0.0), ("hair_depth.png", True, 0, 2, 1.0))
    for i, view in enumerate(camera.keys()):
        os.makedirs(os.path.join(save_root, view), exist_ok=True)     # (cv2.imwrite fails silently without it)
        for name, strands_on, copt, dopt, clear in passes:
            img = r.render(recs[i], H, W, copt, dopt, clear, draw_strands=strands_on, pixel_center=pixel_center)
            _save_png(os.path.join(save_root, view, name), img.cpu().numpy(), swap=name in ("undirectional_map.png", "mask.png"))
