"""Seeded synthetic PMVO scenes: a ring of pinhole cameras looking
at an analytic sphere whose surface carries a meridian tangent field.

Everything is computed in float64 from IEEE basic operations only (+ - * / sqrt,
integer hashing for the noise), so the same arrays come out on every host and on
the GPU (torch CPU and torch ROCm agree bit-for-bit on those ops); camera poses are
the only place sin/cos enter and they are tiny, so fixtures store them explicitly.

Conventions (reference: /root/reference/Utils/Camera_utils.py:19-58,
PMVO.py:378-397, Utils/Render_utils.py:338-340):
  * camera looks down -z; pose in cam_params.json is camera-to-world (c2w);
  * ndc:  u = fx*x/z + cx,  v = fy*y/z + cy  (z < 0 in front of the camera);
  * pixel: col = (-u+1)/2*W,  row = (v+1)/2*H  (image_size = [H, W]);
  * depth map value = (-z_cam/2)*255, background 255, replicated to 3 channels;
  * Ori[..., 0] is the row (down) component, Ori[..., 1] the column component.
"""
import math

import numpy as np
import torch

SPHERE_R = 0.12
BBOX_MIN = (-0.32, -0.32, -0.24)


def make_cameras(V, H, W, radius=0.8, scale=1.7, rings=1, elev_deg=20.0):
    """Return a cam_params.json-style list: {'file','pose' (c2w 4x4),'ndc_prj'[fx,fy,cx,cy]}."""
    fy = 1.7406 * scale
    fx = fy * float(H) / float(W)
    cams = []
    for i in range(V):
        ring = i % rings
        a = 2.0 * math.pi * (i // rings) / max(1, (V + rings - 1) // rings)
        e = 0.0 if rings == 1 else math.radians(elev_deg) * (ring - (rings - 1) / 2.0)
        ca, sa, ce, se = math.cos(a), math.sin(a), math.cos(e), math.sin(e)
        Ry = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]], dtype=np.float64)
        Rx = np.array([[1, 0, 0], [0, ce, -se], [0, se, ce]], dtype=np.float64)
        R = Ry @ Rx
        c2w = np.eye(4, dtype=np.float64)
        c2w[:3, :3] = R
        c2w[:3, 3] = R @ np.array([0.0, 0.0, radius])
        cams.append(dict(file="view_%03d" % i, pose=c2w.tolist(), ndc_prj=[fx, fy, 0.0, 0.0]))
    return cams


def _hash01(view, rows, cols, seed):
    """Deterministic U[0,1) noise per (view,row,col): 32-bit mix done in int64 (no overflow)."""
    x = (rows * 73856093 + cols * 19349663 + (view + 1) * 83492791 + seed * 2654435) & 0xFFFFFFFF
    for mul in (0x7FEB352D, 0x3243F6A9):
        x = x ^ (x >> 15)
        x = (x * mul) & 0xFFFFFFFF
    x = x ^ (x >> 16)
    return x.to(torch.float64) / 4294967296.0


def _project(px, py, pz, w2c, fx, fy, H, W):
    xc = w2c[0][0] * px + w2c[0][1] * py + w2c[0][2] * pz + w2c[0][3]
    yc = w2c[1][0] * px + w2c[1][1] * py + w2c[1][2] * pz + w2c[1][3]
    zc = w2c[2][0] * px + w2c[2][1] * py + w2c[2][2] * pz + w2c[2][3]
    row = (fy * yc / zc + 1.0) / 2.0 * H
    col = (-fx * xc / zc + 1.0) / 2.0 * W
    return row, col, zc


def render_view(cam, view_index, H, W, device="cpu", seed=0, sphere_r=SPHERE_R, quantize=False):
    """Analytic maps of one view -> (depth[H,W], ori[H,W,2], conf[H,W], mask[H,W]) float32 tensors.

    quantize=True pushes orientation/confidence through the reference's 8-bit file
    hand-off (GaborFilter.py:209-210 -> PMVO_utils.py:265-272): integer degrees and
    conf/255, which is what real captures look like.
    """
    c2w = [[float(x) for x in r] for r in cam["pose"]]
    w2c = np.linalg.inv(np.array(c2w, dtype=np.float64)).tolist()
    fx, fy = float(cam["ndc_prj"][0]), float(cam["ndc_prj"][1])
    f64 = dict(dtype=torch.float64, device=device)
    rows = torch.arange(H, device=device, dtype=torch.int64)[:, None].expand(H, W)
    cols = torch.arange(W, device=device, dtype=torch.int64)[None, :].expand(H, W)
    r = rows.to(torch.float64)
    c = cols.to(torch.float64)
    # pixel centre -> camera-space ray (z<0 forward)
    xz = -(2.0 * c / W - 1.0) / fx
    yz = (2.0 * r / H - 1.0) / fy
    inv = 1.0 / torch.sqrt(xz * xz + yz * yz + 1.0)
    dcx, dcy, dcz = -xz * inv, -yz * inv, -inv
    dx = c2w[0][0] * dcx + c2w[0][1] * dcy + c2w[0][2] * dcz
    dy = c2w[1][0] * dcx + c2w[1][1] * dcy + c2w[1][2] * dcz
    dz = c2w[2][0] * dcx + c2w[2][1] * dcy + c2w[2][2] * dcz
    ox, oy, oz = c2w[0][3], c2w[1][3], c2w[2][3]
    b = ox * dx + oy * dy + oz * dz
    cc = ox * ox + oy * oy + oz * oz - sphere_r * sphere_r
    disc = b * b - cc
    hit = disc > 0.0
    t = -b - torch.sqrt(torch.clamp(disc, min=0.0))
    hit = hit & (t > 0.0)
    px, py, pz = ox + t * dx, oy + t * dy, oz + t * dz
    zc = t * dcz
    depth = torch.where(hit, (-zc / 2.0) * 255.0, torch.full_like(zc, 255.0))
    nx, ny, nz = px / sphere_r, py / sphere_r, pz / sphere_r
    # meridian tangent  t = normalize(-e_y + (n.e_y) n)
    tx, ty, tz = ny * nx, -1.0 + ny * ny, ny * nz
    tn = torch.sqrt(tx * tx + ty * ty + tz * tz)
    tn = torch.where(tn > 1e-12, tn, torch.ones_like(tn))
    tx, ty, tz = tx / tn, ty / tn, tz / tn
    r0, c0, _ = _project(px, py, pz, w2c, fx, fy, H, W)
    r1, c1, _ = _project(px + 1e-4 * tx, py + 1e-4 * ty, pz + 1e-4 * tz, w2c, fx, fy, H, W)
    drow, dcol = r1 - r0, c1 - c0
    dn = torch.sqrt(drow * drow + dcol * dcol)
    ok = dn > 1e-12
    dn = torch.where(ok, dn, torch.ones_like(dn))
    drow = torch.where(ok, drow / dn, torch.ones_like(dn))
    dcol = torch.where(ok, dcol / dn, torch.zeros_like(dn))
    facing = torch.clamp(-(nx * dx + ny * dy + nz * dz), min=0.0)
    u = _hash01(view_index, rows, cols, seed)
    conf = torch.where(hit, facing * (1.0 - 0.05 * u), torch.zeros_like(facing))
    drow = torch.where(hit, drow, torch.zeros_like(drow))
    dcol = torch.where(hit, dcol, torch.ones_like(dcol))
    if quantize:
        # Gabor index k <-> line direction (row,col) = (-sin th_k, cos th_k); file pixel = k degrees
        ang = torch.atan2(-drow, dcol) * (180.0 / math.pi)
        k = torch.remainder(torch.round(ang), 180.0)
        thp = (180.0 - k) / 180.0 * math.pi
        drow, dcol = torch.sin(thp), torch.cos(thp)
        conf = torch.floor(conf * 255.0 + 0.5) / 255.0
    ori = torch.stack([drow, dcol], dim=-1)
    mask = hit.to(torch.float64)
    return (depth.to(torch.float32), ori.to(torch.float32), conf.to(torch.float32), mask.to(torch.float32))


def make_scene(V, H, W, device="cpu", seed=0, scale=1.7, rings=1, quantize=False):
    """Compact planes of all views: dict(cams, depth[V,H,W], ori[V,H,W,2], conf[V,H,W], mask[V,H,W])."""
    cams = make_cameras(V, H, W, scale=scale, rings=rings)
    depth = torch.empty((V, H, W), dtype=torch.float32, device=device)
    ori = torch.empty((V, H, W, 2), dtype=torch.float32, device=device)
    conf = torch.empty((V, H, W), dtype=torch.float32, device=device)
    mask = torch.empty((V, H, W), dtype=torch.float32, device=device)
    for i, cam in enumerate(cams):
        d, o, c, m = render_view(cam, i, H, W, device=device, seed=seed, quantize=quantize)
        depth[i], ori[i], conf[i], mask[i] = d, o, c, m
    return dict(cams=cams, depth=depth, ori=ori, conf=conf, mask=mask, image_size=[H, W])


def make_scene_codes(V, H, W, device="cpu", seed=0, scale=1.7, rings=1):
    """The same views as their 8-bit FILE CODES (what a capture's best_ori/, conf/ and hair_mask/ images hold, SURVEY.md
    Appendix A.18): dict(cams, depth [V,H,W] float32, ori_u8 / conf_u8 / mask_u8 [V,H,W] uint8) for PMVO.from_u8.
    Codes as write_capture() below writes them: orientation in integer degrees, floor(conf*255 + 0.5), mask 0 / 255."""
    cams = make_cameras(V, H, W, scale=scale, rings=rings)
    depth = torch.empty((V, H, W), dtype=torch.float32, device=device)
    k8 = torch.empty((V, H, W), dtype=torch.uint8, device=device)
    c8 = torch.empty((V, H, W), dtype=torch.uint8, device=device)
    m8 = torch.empty((V, H, W), dtype=torch.uint8, device=device)
    for i, cam in enumerate(cams):
        d, o, c, m = render_view(cam, i, H, W, device=device, seed=seed, quantize=False)
        ang = torch.atan2(-o[..., 0].double(), o[..., 1].double()) * (180.0 / math.pi)
        depth[i] = d
        k8[i] = torch.remainder(torch.round(ang), 180.0).to(torch.uint8)
        c8[i] = torch.floor(c.double() * 255.0 + 0.5).clamp(0, 255).to(torch.uint8)
        m8[i] = (m * 255).to(torch.uint8)
    return dict(cams=cams, depth=depth, ori_u8=k8, conf_u8=c8, mask_u8=m8, image_size=[H, W])


def scene_to_reference_dicts(scene):
    """Reference-layout host dicts as PMVO.__init__ takes them (PMVO.py:14-26):
    depths[k] [H,W,3], Ori[k] [H,W,2], Conf[k] [H,W], masks[k] [H,W,3] (numpy)."""
    depths, Ori, Conf, masks = {}, {}, {}, {}
    for i, cam in enumerate(scene["cams"]):
        k = cam["file"]
        d = scene["depth"][i].cpu().numpy()
        m = scene["mask"][i].cpu().numpy()
        depths[k] = np.repeat(d[..., None], 3, axis=-1)
        masks[k] = np.repeat(m[..., None], 3, axis=-1).astype(np.float64)
        Ori[k] = scene["ori"][i].cpu().numpy().astype(np.float64)
        Conf[k] = scene["conf"][i].cpu().numpy().astype(np.float64)
    return depths, Ori, Conf, masks


def candidate_points(res=256, seed=0, n_per_voxel=4, sphere_r=SPHERE_R, limit=None):
    """Candidate 3D points: surface voxels of the sphere on the 2x-resolution grid times
    n_per_voxel uniform jitters (SamplePointsAroundmesh semantics,
    /root/reference/Utils/PMVO_utils.py:316-339: np.nonzero order, copies concatenated,
    y and z negated on the way in and out)."""
    vsize = 0.64 / res / 2.0
    g = np.array([2 * res, 2 * res, int(2 * res * 0.75)])
    bbox = np.array(BBOX_MIN, dtype=np.float64)
    # only scan the slab of voxels near the sphere
    lo = np.maximum(np.floor((-(sphere_r + 2 * vsize) - bbox) / vsize).astype(int), 0)
    hi = np.minimum(np.ceil(((sphere_r + 2 * vsize) - bbox) / vsize).astype(int) + 1, g)
    ix = np.arange(lo[0], hi[0])
    iy = np.arange(lo[1], hi[1])
    iz = np.arange(lo[2], hi[2])
    X, Y, Z = np.meshgrid(ix, iy, iz, indexing="ij")
    cx = (X + 0.5) * vsize + bbox[0]
    cy = (Y + 0.5) * vsize + bbox[1]
    cz = (Z + 0.5) * vsize + bbox[2]
    d = np.sqrt(cx * cx + cy * cy + cz * cz)
    sel = np.abs(d - sphere_r) <= 0.5 * vsize
    idx = np.stack([X[sel], Y[sel], Z[sel]], axis=1)  # ij-order == np.nonzero order
    base = np.concatenate([idx] * n_per_voxel, axis=0).astype(np.float64)
    rng = np.random.default_rng(seed)
    sample = (base + rng.random(base.shape)) * vsize + bbox
    sample[:, 1:] *= -1
    if limit is not None:
        sample = sample[:limit]
    return sample


def write_case(root, case="synthetic_sphere", V=24, H=480, W=270, seed=0, scale=1.7, rings=1, res=64):
    """Write a complete on-disk capture in the reference's layout (SURVEY.md §8b "files in") so that
    `python PMVO.py --yaml=configs/reconstruct/<case>` runs through the real loaders:
      ours/cam_params.json, capture_images/<view>.png, render_depth/<view>.npy [H,W,3] f32,
      best_ori/<view>.png (u8 degrees), conf/<view>.png (u8), hair_mask/<view>.png (BGR u8),
      ours/colmap_points.obj (the sphere), ours/bust_long_tsfm.obj, ours/scalp_tsfm.obj."""
    import json
    import os

    from PIL import Image

    base = os.path.join(root, case)
    for d in ("ours", "capture_images", "render_depth", "best_ori", "conf", "hair_mask"):
        os.makedirs(os.path.join(base, d), exist_ok=True)
    cams = make_cameras(V, H, W, scale=scale, rings=rings)
    with open(os.path.join(base, "ours", "cam_params.json"), "w") as f:
        json.dump({"cam_list": cams}, f)
    for i, cam in enumerate(cams):
        d, o, c, m = render_view(cam, i, H, W, seed=seed, quantize=False)
        ang = torch.atan2(-o[..., 0].double(), o[..., 1].double()) * (180.0 / math.pi)
        k = torch.remainder(torch.round(ang), 180.0).to(torch.uint8).numpy()
        c8 = torch.floor(c.double() * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).numpy()
        m8 = (m.numpy() * 255).astype(np.uint8)
        name = cam["file"]
        np.save(os.path.join(base, "render_depth", name + ".npy"), np.repeat(d.numpy()[..., None], 3, axis=2))
        Image.fromarray(k).save(os.path.join(base, "best_ori", name + ".png"))
        Image.fromarray(c8).save(os.path.join(base, "conf", name + ".png"))
        Image.fromarray(np.repeat(m8[..., None], 3, axis=2)).save(os.path.join(base, "hair_mask", name + ".png"))
        Image.fromarray(c8).save(os.path.join(base, "capture_images", name + ".png"))

    def sphere_obj(path, radius, n_lat=48, n_lon=96, y_min=None):
        vs, fs = [], []
        for a in range(n_lat + 1):
            th = math.pi * a / n_lat
            for b in range(n_lon):
                ph = 2 * math.pi * b / n_lon
                vs.append((radius * math.sin(th) * math.cos(ph), radius * math.cos(th), radius * math.sin(th) * math.sin(ph)))
        for a in range(n_lat):
            for b in range(n_lon):
                p00, p01 = a * n_lon + b, a * n_lon + (b + 1) % n_lon
                p10, p11 = p00 + n_lon, p01 + n_lon
                fs.append((p00, p10, p11))
                fs.append((p00, p11, p01))
        if y_min is not None:
            keep = {i for i, v in enumerate(vs) if v[1] >= y_min}
            fs = [t for t in fs if all(i in keep for i in t)]
        with open(path, "w") as f:
            for v in vs:
                f.write("v %.9f %.9f %.9f\n" % v)
            for t in fs:
                f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))

    sphere_obj(os.path.join(base, "ours", "colmap_points.obj"), SPHERE_R, n_lat=4 * res // 8, n_lon=8 * res // 8)
    sphere_obj(os.path.join(base, "ours", "bust_long_tsfm.obj"), 0.09, 24, 48)
    sphere_obj(os.path.join(base, "ours", "scalp_tsfm.obj"), 0.10, 24, 48, y_min=0.03)
    return base
