"""Lossless packed container for the per-view PMVO inputs.

The reference keeps each view's maps as four files (best_ori/<view> and conf/<view> gray 8-bit images, hair_mask/<view>
BGR 8-bit, render_depth/<view>.npy float32 [H,W,3]; Utils/PMVO_utils.py:255-313) and decodes them to float64 on the
host.  The pack stores exactly the information PMVO reads -- the three 8-bit pixel-code planes and channel 0 of the
depth -- in one memory-mappable file, 7 bytes per pixel instead of 36:

    bytes 0..7    magic  b"MHMAPS1\\n"
    bytes 8..15   little-endian u64: length L of the JSON header
    bytes 16..    JSON {"views": [...], "H":, "W":, "planes": {name: {"dtype":, "offset":, "shape":}}}
    then (64-byte aligned, offsets relative to the file start)
                  ori  u8 [V,H,W] | conf u8 [V,H,W] | mask u8 [V,H,W] | depth f32 [V,H,W]

The reference files stay the default input; `data.maps_pack` in the YAML (or --data.maps_pack=) names a pack, which
PMVO.py writes on first use and memory-maps afterwards.  Decoding is PMVO.from_u8's table lookup on the GPU.
"""
import json
import os

import numpy as np

MAGIC = b"MHMAPS1\n"
PLANES = (("ori", np.uint8), ("conf", np.uint8), ("mask", np.uint8), ("depth", np.float32))


def _stack(maps, views, dtype):
    if isinstance(maps, dict):
        return np.stack([np.asarray(maps[v], dtype=dtype) for v in views])
    return np.ascontiguousarray(maps, dtype=dtype)


def write_pack(path, views, ori_u8, conf_u8, mask_u8, depth):
    """views: list of view names; maps: dict view -> [H,W] (or [V,H,W] arrays in `views` order)."""
    views = [str(v) for v in views]
    arrs = {"ori": _stack(ori_u8, views, np.uint8), "conf": _stack(conf_u8, views, np.uint8),
            "mask": _stack(mask_u8, views, np.uint8), "depth": _stack(depth, views, np.float32)}
    V, H, W = arrs["ori"].shape
    for name, a in arrs.items():
        if a.shape != (V, H, W) or V != len(views):
            raise ValueError("maps pack: plane %r has shape %s, expected %s" % (name, a.shape, (len(views), H, W)))
    hdr = {"views": views, "H": H, "W": W, "planes": {}}
    # two passes: the header length decides the first offset
    for _ in range(2):
        blob = json.dumps(hdr).encode()
        off = (16 + len(blob) + 256 + 63) // 64 * 64      # slack so that the second pass cannot move the planes
        for name, dt in PLANES:
            hdr["planes"][name] = {"dtype": np.dtype(dt).str, "offset": off, "shape": [V, H, W]}
            off = (off + arrs[name].nbytes + 63) // 64 * 64
    blob = json.dumps(hdr).encode()
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(blob)).tobytes())
        f.write(blob)
        for name, _ in PLANES:
            f.seek(hdr["planes"][name]["offset"])
            arrs[name].tofile(f)
    os.replace(tmp, path)        # atomic: a concurrent reader sees either no pack or a complete one
    return path


def read_pack(path, views=None):
    """-> dict(views, H, W, ori, conf, mask, depth) with read-only memory maps [V,H,W].  `views`: optional list of
    view names to select (in that order); a view missing from the pack raises KeyError."""
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("%s is not a MonoHair maps pack" % path)
        n = int(np.frombuffer(f.read(8), dtype="<u8")[0])
        hdr = json.loads(f.read(n).decode())
    out = {"views": hdr["views"], "H": hdr["H"], "W": hdr["W"]}
    for name, _ in PLANES:
        d = hdr["planes"][name]
        out[name] = np.memmap(path, mode="r", dtype=np.dtype(d["dtype"]), offset=d["offset"], shape=tuple(d["shape"]))
    if views is not None:
        pos = {v: i for i, v in enumerate(out["views"])}
        missing = [v for v in views if v not in pos]
        if missing:
            raise KeyError("maps pack %s lacks views %s" % (path, missing[:5]))
        sel = [pos[v] for v in views]
        out["views"] = list(views)
        if sel != list(range(len(pos))):
            for name, _ in PLANES:
                out[name] = [out[name][i] for i in sel]
    return out


def pack_case(camera, Ori_path, Conf_path, mask_path, depth_path, out_path, threads=8):
    """Build a pack from the reference's file tree (the inverse is not needed: the files stay where they are)."""
    from .pmvo_utils import load_depth_plane, load_maps_u8

    o, c, m = load_maps_u8(camera, Ori_path, Conf_path, mask_path, threads)
    d = load_depth_plane(camera, depth_path, threads)
    return write_pack(out_path, list(camera.keys()), o, c, m, d)
