"""Golden images from a REAL OpenGL implementation for the two rasterisers (csrc/raster.hip, oracle/raster_oracle.c):
    python tools/gen_golden_gl.py            (build container only; writes tests/golden/gl_raster.npz)
tools/gl_ref/gl_ref.c restates the reference's two GL passes (Utils/Render_utils.py: BustObj triangles, StrandsObj lines,
Renderer state) for OpenGL ES 3.0 and runs them on Google SwiftShader, the software GL that ships inside this image's
`kaleido` package (headless EGL pbuffer).  The reference itself runs on whatever desktop GL driver its machine has; what
every conformant GL shares -- sample positions at pixel centres, one fragment per pixel of a shared edge, perspective-correct
varyings, LESS depth test in draw order, the diamond-exit rule for lines -- is what these fixtures pin; sub-pixel snapping
(SwiftShader: 1/16 pixel, here: 1/256) and interpolation rounding are implementation-defined and covered by tolerances."""
import glob
import math
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import cameras_from_list  # noqa: E402


def find_swiftshader():
    for d in glob.glob("/usr/local/lib/python3*/dist-packages/kaleido/executable/bin/swiftshader"):
        if os.path.exists(os.path.join(d, "libEGL.so")):
            return d
    raise SystemExit("SwiftShader (kaleido/executable/bin/swiftshader) not found")


def build(tmp):
    exe = os.path.join(tmp, "gl_ref")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tools", "gl_ref", "gl_ref.c"), "-ldl"])
    return exe


def run(exe, ss, tmp, W, H, clear, draws, depth_bits=24):
    job, out = os.path.join(tmp, "job.bin"), os.path.join(tmp, "out.bin")
    with open(job, "wb") as f:
        f.write(struct.pack("<ii3fii", W, H, *clear, depth_bits, len(draws)))
        for d in draws:
            pos = np.ascontiguousarray(d["pos"], np.float32)
            nidx = 0 if d["kind"] == 1 else int(np.asarray(d["idx"]).size)
            f.write(struct.pack("<iiiif", d["kind"], len(pos), nidx, d["option"], d.get("width", 1.0)))
            f.write(np.ascontiguousarray(d["proj"], np.float32).tobytes())
            f.write(np.ascontiguousarray(d["pose"], np.float32).tobytes())
            f.write(pos.tobytes())
            if d["kind"] == 1:
                f.write(np.ascontiguousarray(d["tan"], np.float32).tobytes())
            else:
                f.write(np.ascontiguousarray(d["idx"], np.uint32).tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=ss)
    subprocess.check_call([exe, os.path.join(ss, "libEGL.so"), os.path.join(ss, "libGLESv2.so"), job, out], env=env)
    raw = np.fromfile(out, np.float32)
    lw = raw[:2].copy()
    img = raw[2:].reshape(H, W, 4)
    return np.flip(img, 0)[..., :3].copy(), lw          # Renderer.ReadBuffer: 3 components, flipped to a top-left origin


def uv_sphere(radius, n_lat, n_lon, centre=(0, 0, 0)):
    vs, fs = [], []
    for a in range(n_lat + 1):
        th = math.pi * a / n_lat
        for b in range(n_lon):
            ph = 2 * math.pi * b / n_lon
            vs.append((radius * math.sin(th) * math.cos(ph) + centre[0], radius * math.cos(th) + centre[1],
                       radius * math.sin(th) * math.sin(ph) + centre[2]))
    for a in range(n_lat):
        for b in range(n_lon):
            p00, p01 = a * n_lon + b, a * n_lon + (b + 1) % n_lon
            p10, p11 = p00 + n_lon, p01 + n_lon
            fs += [(p00, p10, p11), (p00, p11, p01)]
    return np.array(vs, np.float32), np.array(fs, np.int32)


def strands_on_sphere(rng, n, radius):
    out = []
    for _ in range(n):
        m = int(rng.integers(8, 40))
        th0, ph0 = rng.uniform(0.3, 2.6), rng.uniform(0, 2 * math.pi)
        dth, dph = rng.normal(0, 0.02), rng.normal(0, 0.03)
        r = radius * (1.0 + rng.uniform(0.01, 0.12))
        k = np.arange(m)
        th, ph = th0 + dth * k, ph0 + dph * k
        out.append(np.stack([r * np.sin(th) * np.cos(ph), r * np.cos(th), r * np.sin(th) * np.sin(ph)], 1).astype(np.float32))
    return out


def main():
    from monohair_amd.render import strand_line_buffers

    ss = find_swiftshader()
    rng = np.random.default_rng(3)
    H, W = 240, 136
    cams = synth.make_cameras(24, H, W, scale=1.7)
    C = cameras_from_list(cams)
    names = list(C.keys())
    out = {"H": H, "W": W, "cam_pose": np.stack([np.asarray(c["pose"], np.float64) for c in cams]),
           "cam_ndc": np.stack([np.asarray(c["ndc_prj"], np.float64) for c in cams]),
           "gl": np.array("Google SwiftShader, OpenGL ES 3.0 (kaleido bundle), 24-bit depth buffer")}
    v1, f1 = uv_sphere(synth.SPHERE_R, 40, 80)
    v2, f2 = uv_sphere(synth.SPHERE_R * 0.6, 24, 48, centre=(0.04, -0.05, 0.03))       # pokes through the first one
    soup_v = rng.uniform(-0.12, 0.12, (300, 3)).astype(np.float32)
    soup_f = rng.integers(0, 300, (120, 3)).astype(np.int32)
    strands = strands_on_sphere(rng, 60, synth.SPHERE_R)
    lp, lt = strand_line_buffers(strands)
    out.update(v1=v1, f1=f1, v2=v2, f2=f2, soup_v=soup_v, soup_f=soup_f, line_pts=lp, line_tan=lt)
    views = [0, 7, 13]
    out["views"] = np.array(views)
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        for vi in views:
            c = C[names[vi]]
            proj, pose = c.proj.cpu().numpy().astype(np.float32), c.pose.cpu().numpy().astype(np.float32)

            def tri(v, f, option=0):
                return dict(kind=0, pos=v, idx=f, option=option, proj=proj, pose=pose)

            def lines(option, width):
                return dict(kind=1, pos=lp, tan=lt, option=option, width=width, proj=proj, pose=pose)

            # render_bust_hair_depth: two meshes, depth colour, white background
            img, lw = run(exe, ss, tmp, W, H, (1, 1, 1), [tri(v1, f1), tri(v2, f2)])
            out["depth_two_meshes_%d" % vi] = img[..., 0]
            img, _ = run(exe, ss, tmp, W, H, (1, 1, 1), [tri(soup_v, soup_f)])
            out["depth_soup_%d" % vi] = img[..., 0]
            # render_data: strands over the bust; undirectional map (option 2 / bust black), mask (3), hair depth (0 / bust white)
            for width in (1.0,):        # this GL's line width range is [1, 1]: a request for 3 (the reference's) is clamped
                tag = "w%d_%d" % (int(width), vi)
                img, _ = run(exe, ss, tmp, W, H, (0, 0, 0), [tri(v1, f1, 1), lines(2, width)])
                out["strand_color_" + tag] = img
                img, _ = run(exe, ss, tmp, W, H, (0, 0, 0), [tri(v1, f1, 1), lines(3, width)])
                out["strand_mask_" + tag] = img[..., 0]
                img, _ = run(exe, ss, tmp, W, H, (1, 1, 1), [tri(v1, f1, 2), lines(0, width)])
                out["strand_depth_" + tag] = img[..., 0]
            out["line_width_range"] = lw
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gl_raster.npz"), **out)
    print("line width range of this GL:", out["line_width_range"])
    for k in sorted(out):
        if k.startswith(("depth_", "strand_")):
            print(k, out[k].shape, float(np.asarray(out[k]).sum()))


if __name__ == "__main__":
    main()
