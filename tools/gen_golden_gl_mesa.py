"""Golden images from a DESKTOP OpenGL implementation (Mesa llvmpipe, OpenGL 4.5 core) for the two rasterisers, with the
reference's own GLSL and its default 3-pixel lines:
    python tools/gen_golden_gl_mesa.py       (build container only; writes tests/golden/gl_mesa.npz)
What round 2 could not pin with SwiftShader (ES 3.0, line widths [1,1], shaders restated for ES): here the four shader
strings are CUT OUT OF /root/reference/Utils/Render_utils.py at generation time (`ast` on the file; nothing of it is stored
in this repository -- the fixture holds images) and compiled as they are by tools/gl_ref/gl_ref_mesa.c, which drives Mesa's
software DRI driver as its own minimal loader (the image has no X server, no EGL and no OSMesa).  Scenes as in
tools/gen_golden_gl.py, lines at widths 1, 2 and 3."""
import ast
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import gen_golden_gl as G  # noqa: E402
from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import cameras_from_list  # noqa: E402

REF_FILE = "/root/reference/Utils/Render_utils.py"


def reference_shaders():
    """{class name: (vertex source, fragment source)} of StrandsObj / BustObj, read from the reference file's syntax tree"""
    tree = ast.parse(open(REF_FILE).read())
    out = {}
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name in ("StrandsObj", "BustObj"):
            for fn in cls.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "loadShader":
                    for node in ast.walk(fn):
                        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "program":
                            kw = {k.arg: k.value.value for k in node.keywords}
                            out[cls.name] = (kw["vertex_shader"], kw["fragment_shader"])
    assert set(out) == {"StrandsObj", "BustObj"}
    # GLSL wants `#version` on the first line: the strings start with a newline and indentation
    return {k: tuple(s.strip() + "\n" for s in v) for k, v in out.items()}


def build(tmp):
    exe = os.path.join(tmp, "gl_ref_mesa")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tools", "gl_ref", "gl_ref_mesa.c"), "-ldl"])
    return exe


def run(exe, shaders_txt, tmp, W, H, clear, draws, depth_bits=24):
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
            f.write(np.ascontiguousarray(d["tan"], np.float32).tobytes() if d["kind"] == 1 else
                    np.ascontiguousarray(d["idx"], np.uint32).tobytes())
    subprocess.check_call([exe, job, shaders_txt, out], stderr=subprocess.DEVNULL)
    raw = np.fromfile(out, np.float32)
    return np.flip(raw[2:].reshape(H, W, 4), 0)[..., :3].copy(), raw[:2].copy()


def main():
    from monohair_amd.render import strand_line_buffers

    sh = reference_shaders()
    rng = np.random.default_rng(3)
    H, W = 240, 136
    cams = synth.make_cameras(24, H, W, scale=1.7)
    C = cameras_from_list(cams)
    names = list(C.keys())
    out = {"H": H, "W": W, "cam_pose": np.stack([np.asarray(c["pose"], np.float64) for c in cams]),
           "cam_ndc": np.stack([np.asarray(c["ndc_prj"], np.float64) for c in cams]),
           "gl": np.array("Mesa 23.2.1 llvmpipe, OpenGL 4.5 core / GLSL 3.30 shaders of the reference, 24-bit depth buffer")}
    v1, f1 = G.uv_sphere(synth.SPHERE_R, 40, 80)
    v2, f2 = G.uv_sphere(synth.SPHERE_R * 0.6, 24, 48, centre=(0.04, -0.05, 0.03))
    soup_v = rng.uniform(-0.12, 0.12, (300, 3)).astype(np.float32)
    soup_f = rng.integers(0, 300, (120, 3)).astype(np.int32)
    strands = G.strands_on_sphere(rng, 60, synth.SPHERE_R)
    lp, lt = strand_line_buffers(strands)
    out.update(v1=v1, f1=f1, v2=v2, f2=f2, soup_v=soup_v, soup_f=soup_f, line_pts=lp, line_tan=lt)
    views = [0, 7, 13]
    out["views"] = np.array(views)
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        stxt = os.path.join(tmp, "shaders.txt")
        with open(stxt, "w") as f:
            f.write("\n=====\n".join([sh["BustObj"][0], sh["BustObj"][1], sh["StrandsObj"][0], sh["StrandsObj"][1]]))
        for vi in views:
            c = C[names[vi]]
            proj, pose = c.proj.cpu().numpy().astype(np.float32), c.pose.cpu().numpy().astype(np.float32)

            def tri(v, f, option=0):
                return dict(kind=0, pos=v, idx=f, option=option, proj=proj, pose=pose)

            def lines(option, width):
                return dict(kind=1, pos=lp, tan=lt, option=option, width=width, proj=proj, pose=pose)

            img, lw = run(exe, stxt, tmp, W, H, (1, 1, 1), [tri(v1, f1), tri(v2, f2)])
            out["depth_two_meshes_%d" % vi] = img[..., 0]
            img, _ = run(exe, stxt, tmp, W, H, (1, 1, 1), [tri(soup_v, soup_f)])
            out["depth_soup_%d" % vi] = img[..., 0]
            for width in (1.0, 2.0, 3.0):      # 3 is the reference's (Render_utils.py:28)
                tag = "w%d_%d" % (int(width), vi)
                img, _ = run(exe, stxt, tmp, W, H, (0, 0, 0), [tri(v1, f1, 1), lines(2, width)])
                out["strand_color_" + tag] = img
                img, _ = run(exe, stxt, tmp, W, H, (0, 0, 0), [tri(v1, f1, 1), lines(3, width)])
                out["strand_mask_" + tag] = img[..., 0]
                img, _ = run(exe, stxt, tmp, W, H, (1, 1, 1), [tri(v1, f1, 2), lines(0, width)])
                out["strand_depth_" + tag] = img[..., 0]
                # strands alone (nothing hides them): the pure line rule
                img, _ = run(exe, stxt, tmp, W, H, (0, 0, 0), [lines(3, width)])
                out["strand_alone_" + tag] = img[..., 0]
            out["line_width_range"] = lw
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gl_mesa.npz"), **out)
    print("line width range of this GL:", out["line_width_range"])
    for k in sorted(out):
        if k.startswith(("depth_", "strand_")):
            print(k, out[k].shape, float(np.asarray(out[k]).sum()))


if __name__ == "__main__":
    main()
