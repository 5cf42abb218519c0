"""Pinhole camera of the PMVO path -- host-side mirror of the reference's
`Utils/Camera_utils.py` (Camera :10-36, load_cam :141-146, parsing_camera :148-163).

The device never sees this class: cameras are flattened into 48-float records
(`CAM_STRIDE`) that `mh_ctx_set_views` uploads once:
    [0:16]  pose   world->camera 4x4, row-major        (Camera.pose)
    [16:32] proj   GL-style projection 4x4, row-major   (Camera.get_projection_matrix :19-36)
    [32:41] rinv   inverse of pose[:3,:3], row-major    (torch.linalg.inv at Camera_utils.py:104)
The 3x3 inverse is taken with torch.linalg.inv on the CPU in float32, i.e. by the very
call the reference makes, so the unprojection kernel multiplies by the same bits.
"""
import json
import os

import numpy as np
import torch

CAM_STRIDE = 48
ZFAR = 100.0
ZNEAR = 0.1


class Camera:
    def __init__(self, proj, pose, id, to_tensor=True):
        self.id = id
        self.proj = torch.from_numpy(self.get_projection_matrix(*[float(x) for x in proj])).type(torch.float)
        self.pose = torch.from_numpy(np.asarray(pose, dtype=np.float64)).type(torch.float)

    def get_projection_matrix(self, fx, fy, cx, cy):
        """Camera_utils.py:19-36: GL-style projection, near 0.1, far 100, camera looking down -z"""
        return np.array(
            [
                [fx, 0, cx, 0],
                [0, fy, cy, 0],
                [0, 0, (-ZFAR - ZNEAR) / (ZFAR - ZNEAR), -2.0 * ZFAR * ZNEAR / (ZFAR - ZNEAR)],
                [0, 0, -1, 0],
            ],
            dtype=np.float64,
        )

    # ---- tensor utilities of the reference's class (Camera_utils.py:38-139), plain torch on whatever device the
    # ---- argument lives on.  The PMVO kernels do NOT call these: they evaluate the same formulas from the records.
    def projection(self, vertices, debug=False):
        """[N,3] world points -> (uv [N,2] = (proj @ pose @ X)[:2] / z_cam, z_cam [N])  (Camera_utils.py:38-58)"""
        self.proj, self.pose = self.proj.to(vertices.device), self.pose.to(vertices.device)
        hom = torch.cat([vertices.permute(1, 0), torch.ones((1, vertices.size(0)), device=vertices.device)])
        cam = torch.matmul(self.pose, hom)
        z = cam[2:3, :]
        uv = torch.matmul(self.proj, cam)
        uv[:2] /= z
        return uv.transpose(1, 0)[:, :2], z[0]

    @staticmethod
    def _extent(image_size, device):
        """(W, H) as a float row: ndc x runs along the image width, ndc y along its height"""
        H, W = float(image_size[0]), float(image_size[1])
        return torch.tensor([W, H], device=device, dtype=torch.float)

    def uv2pixel(self, uv, image_size, device):
        """ndc [-1,1]^2 (x to the left, y down) -> unrounded pixel position as (row, col).  Same arithmetic, operation
        by operation, as Camera_utils.py:60-71 -- negate x, (. + 1) / 2, times (W, H) -- and, as there, the scaled
        (col, row) values are left behind in the argument."""
        col_row = torch.stack([uv[:, 0] * -1, uv[:, 1]], dim=1)
        col_row = (col_row + 1) / 2
        col_row = col_row * self._extent(image_size, device)
        uv[:, :2] = col_row
        return torch.stack([uv[:, 1], uv[:, 0]], dim=1)

    def pixel2uv(self, uv, image_size, device):
        """inverse of uv2pixel: (row, col) pixel -> ndc (Camera_utils.py:73-78: / (W, H), * 2 - 1, negate x)"""
        col_row = torch.stack([uv[:, 1], uv[:, 0]], dim=1) / self._extent(image_size, device)
        ndc = col_row * 2 - 1
        return torch.stack([-ndc[:, 0], ndc[:, 1]], dim=1)

    def reprojection(self, uv, z, to_world=False):
        """ndc + camera depth -> camera-space homogeneous points [N,4], or world points [N,3] (Camera_utils.py:81-109)"""
        self.proj, self.pose = self.proj.to(uv.device), self.pose.to(uv.device)
        cam = torch.ones((4, uv.size(0)), dtype=uv.dtype, device=uv.device)
        cam[0] = (uv[:, 0] - self.proj[0, 2]) / self.proj[0, 0] * z
        cam[1] = (uv[:, 1] - self.proj[1, 2]) / self.proj[1, 1] * z
        cam[2] = z
        if not to_world:
            return cam.permute(1, 0)
        world = torch.matmul(torch.linalg.inv(self.pose[:3, :3]), cam[:3] - self.pose[:3, 3:4])
        return world.permute(1, 0)

    def camera2world(self, points):
        """camera-space [N,3] -> world homogeneous [N,4] (Camera_utils.py:111-116)"""
        self.pose = self.pose.to(points.device)
        hom = torch.cat([points, torch.ones((points.size(0), 1), device=points.device)], 1).permute(1, 0)
        return torch.matmul(torch.linalg.inv(self.pose), hom).permute(1, 0)

    def record(self):
        rec = np.zeros(CAM_STRIDE, dtype=np.float32)
        rec[0:16] = self.pose.cpu().numpy().reshape(-1)
        rec[16:32] = self.proj.cpu().numpy().reshape(-1)
        rec[32:41] = torch.linalg.inv(self.pose[:3, :3].cpu()).numpy().reshape(-1)
        return rec


def camera_records(cameras):
    """dict/list of Camera -> [V, CAM_STRIDE] float32, in insertion order (= view order, PMVO.py:21-28)."""
    cams = list(cameras.values()) if isinstance(cameras, dict) else list(cameras)
    return np.stack([c.record() for c in cams], axis=0)


def cameras_from_list(cam_list):
    """cam_params.json entries {'file','pose' (c2w),'ndc_prj'} -> ordered dict of Camera (pose inverted)."""
    out = {}
    for c in cam_list:
        out[c["file"]] = Camera(c["ndc_prj"], np.linalg.inv(np.array(c["pose"])), c["file"])
    return out


def load_cam(path):
    with open(path, "r") as f:
        return json.load(f)["cam_list"]


def parsing_camera(cam, image_path=None):
    """View subsampling of Camera_utils.py:148-163: stride 4 above 500 capture images, 2 above 300.
    (The reference's membership test ends in an always-true `or c['file']+'.jpg'`, so every strided
    camera is kept; we keep that behaviour.)"""
    step = 1
    if image_path is not None:
        n = len(os.listdir(image_path))
        if n > 500:
            step = 4
        elif n > 300:
            step = 2
    return cameras_from_list(cam[::step])
