"""Golden vectors for the small helpers of Utils/PMVO_utils.py that sit on the file boundary of the path (voxel <-> world
:407-420, the .mat readers :86-113, the .hair reader / writers :47-83,662-680), from the imported reference:
    python tools/gen_golden_utils.py          (build container only; writes tests/golden/utils_small.npz)"""
import os
import sys
import tempfile

import numpy as np
import scipy.io
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(1, ROOT)
from ref_import import import_reference  # noqa: E402


def main():
    R = import_reference()
    U = R["PMVO_utils"]
    rng = np.random.default_rng(17)
    out = {}
    vox = torch.from_numpy(rng.uniform(-5, 260, (200, 3)).astype(np.float32))
    out["vox_in"] = vox.numpy().copy()
    out["vox_to_points"] = U.voxel_to_points(vox.clone()).numpy()
    pts = torch.from_numpy(rng.uniform(-0.4, 0.4, (200, 3)).astype(np.float32))
    out["pts_in"] = pts.numpy().copy()
    out["points_to_voxel"] = U.points_to_voxel(pts.clone()).numpy()
    with tempfile.TemporaryDirectory() as d:
        X, Y, Z = 6, 5, 4
        occ = (rng.random((Y, X, Z)) > 0.6).astype(np.float64)                     # Occ [Y,X,Z]
        ori = rng.normal(size=(Y, X, 3 * Z))                                       # Ori [Y,X,3*Z]
        scipy.io.savemat(os.path.join(d, "Occ3D.mat"), {"Occ": occ})
        scipy.io.savemat(os.path.join(d, "Ori3D.mat"), {"Ori": ori})
        out["mat_occ"], out["mat_ori"] = occ, ori
        for flip in (False, True):
            out["occ_flip%d" % flip] = U.get_ground_truth_3D_occ(os.path.join(d, "Occ3D.mat"), flip=flip)
            out["ori_flip%d" % flip] = U.get_ground_truth_3D_ori(os.path.join(d, "Ori3D.mat"), flip=flip)
        strands = [rng.normal(size=(int(n), 3)).astype(np.float32) for n in (2, 7, 1, 30)]
        bust = np.array([0.006, -1.644, 0.010])
        for k, s in enumerate(strands):
            out["strand%d" % k] = s
        out["bust_to_origin"] = bust
        for translate in (True, False):
            p = os.path.join(d, "s%d.hair" % translate)
            U.save_hair_strands(p, [s.copy() for s in strands], bust, translate=translate)
            out["hair_bytes_t%d" % translate] = np.frombuffer(open(p, "rb").read(), np.uint8)
            seg, pt = U.load_strand(p)
            out["load_seg_t%d" % translate], out["load_pts_t%d" % translate] = np.array(seg), pt
        p = os.path.join(d, "w.hair")
        U.write_strand(np.concatenate(strands, 0), p, [len(s) for s in strands])
        out["write_strand_bytes"] = np.frombuffer(open(p, "rb").read(), np.uint8)
    # Camera tensor utilities (Utils/Camera_utils.py:38-116) of three cameras of a synthetic rig
    from monohair_amd import synth

    H, W = 120, 90
    cams = synth.make_cameras(24, H, W, scale=1.5, rings=2)
    C = R["Camera_utils"]
    P = rng.normal(0, 0.1, (64, 3)).astype(np.float32)
    out["cam_points"] = P
    out["cam_pose_c2w"] = np.stack([np.asarray(c["pose"], np.float64) for c in cams])
    out["cam_ndc"] = np.stack([np.asarray(c["ndc_prj"], np.float64) for c in cams])
    out["cam_views"] = np.array([0, 5, 17])
    for i in (0, 5, 17):
        cam = C.Camera(cams[i]["ndc_prj"], np.linalg.inv(np.array(cams[i]["pose"])), cams[i]["file"])
        uv, zz = cam.projection(torch.from_numpy(P))
        out["cam%d_uv" % i], out["cam%d_z" % i] = uv.numpy(), zz.numpy()
        pix = cam.uv2pixel(uv.clone(), [H, W], "cpu")
        out["cam%d_pix" % i] = pix.numpy()
        out["cam%d_uvback" % i] = cam.pixel2uv(pix.clone(), [H, W], "cpu").numpy()
        out["cam%d_world" % i] = cam.reprojection(uv, zz, to_world=True).numpy()
        camv = cam.reprojection(uv, zz, to_world=False)
        out["cam%d_camv" % i] = camv.numpy()
        out["cam%d_c2w" % i] = cam.camera2world(camv[:, :3]).numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "utils_small.npz"), **out)
    print("written", len(out), "arrays")


if __name__ == "__main__":
    main()
