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
        fx, fy, cx, cy = [float(x) for x in proj]
        mat = np.array(
            [
                [fx, 0, cx, 0],
                [0, fy, cy, 0],
                [0, 0, (-ZFAR - ZNEAR) / (ZFAR - ZNEAR), -2.0 * ZFAR * ZNEAR / (ZFAR - ZNEAR)],
                [0, 0, -1, 0],
            ],
            dtype=np.float64,
        )
        self.id = id
        self.proj = torch.from_numpy(mat).type(torch.float)
        self.pose = torch.from_numpy(np.asarray(pose, dtype=np.float64)).type(torch.float)

    def record(self):
        rec = np.zeros(CAM_STRIDE, dtype=np.float32)
        rec[0:16] = self.pose.numpy().reshape(-1)
        rec[16:32] = self.proj.numpy().reshape(-1)
        rec[32:41] = torch.linalg.inv(self.pose[:3, :3]).numpy().reshape(-1)
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
