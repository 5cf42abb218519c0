"""GPU: mh_render_depth (csrc/raster.hip) against its CPU statement (oracle/raster_oracle.c), bit for bit, and the
rendered depth maps as PMVO input."""
import numpy as np
import pytest
import torch

import oracle
from monohair_amd import synth
from monohair_amd.camera import camera_records, cameras_from_list
from test_raster_host import uv_sphere

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _render(rec, meshes, H, W, pc=0.5, channels=1):
    from monohair_amd.render import DepthRenderer

    return DepthRenderer(meshes, DEV).render(rec, H, W, pc, channels=channels).cpu().numpy()


@pytest.mark.parametrize("seed,H,W,pc", [(0, 96, 72, 0.5), (1, 200, 333, 0.0), (2, 64, 64, 0.5), (3, 270, 480, 0.25)])
def test_random_triangle_soup_bit_exact(seed, H, W, pc):
    rng = np.random.default_rng(seed)
    cams = synth.make_cameras(20, H, W, scale=float(rng.uniform(0.8, 2.0)), rings=2)
    rec = camera_records(cameras_from_list(cams))
    nv, nf = 4000, 9000
    verts = rng.normal(0, 0.12, (nv, 3)).astype(np.float32)
    verts[:50] *= 20                      # some far outside the frustum / behind the camera
    faces = rng.integers(0, nv, (nf, 3)).astype(np.int32)
    near = rng.integers(0, nv - 3, nf // 2)                 # half of the triangles small (neighbouring random verts
    verts2 = verts.copy()                                   # moved close together), half spanning the image
    verts2[near + 1] = verts2[near] + rng.normal(0, 0.004, (len(near), 3)).astype(np.float32)
    verts2[near + 2] = verts2[near] + rng.normal(0, 0.004, (len(near), 3)).astype(np.float32)
    faces[:len(near)] = np.stack([near, near + 1, near + 2], 1)
    faces[-3:] = [[0, 0, 1], [5, 5, 5], [1, 2, nv + 4]]    # degenerate and out-of-range entries
    for v in (0, 7, 13):
        want, cov = oracle.render_depth(rec[v], verts2, faces, H, W, pc)
        got = _render(rec[v], [(verts2, faces)], H, W, pc)
        assert cov > 0.2 * H * W
        assert np.array_equal(got, want)


def test_sphere_channels_meshes_and_empty():
    H, W = 240, 136
    cams = synth.make_cameras(24, H, W, scale=1.7)
    rec = camera_records(cameras_from_list(cams))
    v, f = uv_sphere(synth.SPHERE_R, 192, 384)
    bv, bf = uv_sphere(0.09, 24, 48)
    bv = bv + np.array([0, -0.1, 0], np.float32)
    want, _ = oracle.render_depth(rec[3], np.concatenate([v, bv]), np.concatenate([f, bf + len(v)]), H, W, 0.5, 3)
    got = _render(rec[3], [(v, f), (bv, bf)], H, W, 0.5, channels=3)            # two meshes, the .npy layout
    assert got.shape == (H, W, 3) and np.array_equal(got, want)
    analytic = synth.render_view(cams[3], 3, H, W, seed=0, quantize=False)[0].numpy()
    one = _render(rec[3], [(v, f)], H, W, 0.0)
    both = (analytic < 255) & (one < 255)
    assert np.abs(one[both] - analytic[both]).max() < 0.05
    empty = _render(rec[0], [(np.zeros((0, 3)), np.zeros((0, 3), int))], 32, 48)
    assert empty.shape == (32, 48) and (empty == 255).all()


def test_rendered_depth_feeds_pmvo():
    """The rasterised depth (pixel centres at PMVO's integer positions) is interchangeable with the analytic depth of
    the synthetic scene: the same points are visible and forward() finds the same directions almost everywhere."""
    from monohair_amd.pmvo import PMVO
    from monohair_amd.render import render_depth_planes

    V, H, W = 24, 240, 136
    scene = synth.make_scene(V, H, W, seed=0, quantize=True)
    camera = cameras_from_list(scene["cams"])
    rec = camera_records(camera)
    v, f = uv_sphere(synth.SPHERE_R, 192, 384)
    depth = render_depth_planes(camera, [(v, f)], [H, W], DEV, pixel_center=0.0)
    assert depth.shape == (V, H, W)
    kw = dict(device=DEV, patch_size=3, conf_threshold=0.15)
    a = PMVO.from_planes(rec, scene["depth"].to(DEV), scene["ori"].to(DEV), scene["conf"].to(DEV),
                         scene["mask"].to(DEV), **kw)
    b = PMVO.from_planes(rec, depth, scene["ori"].to(DEV), scene["conf"].to(DEV), scene["mask"].to(DEV), **kw)
    pts = synth.candidate_points(res=32, seed=0)[:2000]
    sa, sb = a.filter_points(pts)[0], b.filter_points(pts)[0]
    assert (sa == sb).float().mean() > 0.97
    keep = (sa & sb).cpu().numpy()
    oa, ob = a.forward(pts[keep])[1], b.forward(pts[keep])[1]
    cosv = (oa * ob).sum(1).abs()
    assert (cosv > 0.999).float().mean() > 0.9


def test_device_resident_hand_off_gabor_raster_pmvo():
    """Images -> Gabor codes (device) + rasterised depth (device) -> PMVO.from_u8 on the device tensors: nothing goes
    through the host or the disk, and the resident maps equal those built from host copies of the same arrays."""
    from monohair_amd.gabor import orientation_maps_device
    from monohair_amd.pmvo import PMVO
    from monohair_amd.render import render_depth_planes

    V, H, W = 20, 160, 120
    cams = synth.make_cameras(V, H, W, scale=1.6)
    camera = cameras_from_list(cams)
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:W]
    images = [np.clip(127 + 80 * np.cos(2 * np.pi * (xx * np.cos(0.1 * i) + yy * np.sin(0.1 * i)) / 4.0)
                      + rng.normal(0, 5, (H, W)), 0, 255).astype(np.uint8) for i in range(V)]
    k8, c8 = orientation_maps_device(images, DEV, return_codes=True)
    assert k8.dtype == torch.uint8 and k8.shape == (V, H, W) and k8.is_cuda
    v, f = uv_sphere(synth.SPHERE_R, 96, 192)
    depth = render_depth_planes(camera, [(v, f)], [H, W], DEV, pixel_center=0.0)
    m8 = (depth < 255).to(torch.uint8) * 255
    kw = dict(device=DEV, image_size=[H, W], patch_size=3, conf_threshold=0.1)
    a = PMVO.from_u8(camera, depth, k8, c8, m8, **kw)
    b = PMVO.from_u8(camera, depth.cpu().numpy(), k8.cpu().numpy(), c8.cpu().numpy(), m8.cpu().numpy(), **kw)
    pts = synth.candidate_points(res=32, seed=1)[:1500]
    for pm in (a, b):
        pm.Compute_Visible_and_Ori(pts)
    for name in ("visible", "Ori", "Conf", "mask", "Ori_patch", "Conf_patch"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert (a.visible > -1).any() and float(a.Conf.max()) > 0.5


# ---------------------------------------------------------------------------------------------------------------------
# strand-segment renderer (mh_render_strands) == its C statement, bit for bit
# ---------------------------------------------------------------------------------------------------------------------
def _random_strands(rng, n, radius=0.125):
    out = []
    for _ in range(n):
        d = rng.normal(size=3)
        p = d / np.linalg.norm(d) * radius * rng.uniform(0.9, 1.15)
        steps = rng.normal(0, 0.0025, size=(int(rng.integers(2, 60)), 3)) + rng.normal(0, 0.002, size=(1, 3))
        out.append(p + np.cumsum(steps, 0))
    return out


@pytest.mark.parametrize("seed,H,W,pc", [(0, 160, 120, 0.5), (1, 333, 200, 0.0), (2, 1280, 720, 0.5)])
def test_strand_renderer_bit_exact_vs_spec(seed, H, W, pc):
    from monohair_amd.render import StrandRenderer, strand_line_buffers

    rng = np.random.default_rng(seed)
    cams = synth.make_cameras(12, H, W, scale=float(rng.uniform(0.8, 1.6)), rings=2)
    rec = camera_records(cameras_from_list(cams))
    bv, bf = uv_sphere(0.11, 48, 96)
    strands = _random_strands(rng, 1500)
    strands.append(np.zeros((1, 3)))                               # a one-point strand has no segment
    strands.append(np.array([[0.2, 0, 0], [0.2, 0, 0]]))           # a zero-length segment (zero tangent)
    strands.append(np.array([[0, 0, 5.0], [0.01, 0, 5.0]]))        # behind some cameras
    lp, lt = strand_line_buffers(strands)
    r = StrandRenderer(strands, bv, bf, DEV)
    assert r.nseg == len(lp) // 2
    for v in (0, 5, 9):
        for copt, dopt, clear, on in ((2, 1, 0.0, True), (3, 1, 0.0, True), (0, 2, 1.0, True), (1, 0, 0.5, True),
                                      (0, 0, 1.0, False)):
            want, prim, owned = oracle.render_strands(rec[v], bv, bf, lp, lt, H, W, pc, 3, copt if on else -1, dopt, clear)
            got = r.render(rec[v], H, W, copt, dopt, clear, draw_strands=on, pixel_center=pc).cpu().numpy()
            assert np.array_equal(got, want), (v, copt, dopt)
            if on:
                assert owned > 0.01 * H * W and (prim >= 0).sum() > owned
        # 1-pixel lines, and the end-pixel-included rule used for the comparison with SwiftShader
        for width, rule in ((1, 0), (1, 1), (3, 1)):
            want, _, _ = oracle.render_strands(rec[v], bv, bf, lp, lt, H, W, pc, width, 2, 1, 0.0, line_rule=rule)
            got = r.render(rec[v], H, W, 2, 1, 0.0, pixel_center=pc, line_width=width, line_rule=rule).cpu().numpy()
            assert np.array_equal(got, want), (v, width, rule)
    # strands only (no mesh) and nothing at all
    r2 = StrandRenderer(strands, np.zeros((0, 3)), np.zeros((0, 3), int), DEV)
    want, prim, _ = oracle.render_strands(rec[1], np.zeros((0, 3)), np.zeros((0, 3), np.int32), lp, lt, H, W, pc, 3, 2, 1, 0.0)
    assert np.array_equal(r2.render(rec[1], H, W, 2, 1, 0.0, pixel_center=pc).cpu().numpy(), want) and (prim >= 0).any()
    r3 = StrandRenderer([], np.zeros((0, 3)), np.zeros((0, 3), int), DEV)
    assert (r3.render(rec[1], 40, 30, 2, 1, 0.25).cpu().numpy() == 0.25).all()


# ---------------------------------------------------------------------------------------------------------------------
# both rasterisers against a REAL OpenGL implementation (Google SwiftShader, tests/golden/gl_raster.npz written by
# tools/gen_golden_gl.py in the build container): same thresholds as the oracle's CPU test (tests/test_raster_host.py)
# ---------------------------------------------------------------------------------------------------------------------
def test_hip_rasterisers_against_real_opengl():
    import os

    from conftest import GOLDEN
    from gl_checks import check_depth_against_gl, check_depth_same_grid, check_strands_against_gl
    from monohair_amd import _lib
    from monohair_amd.pmvo_utils import _ctx_for
    from monohair_amd.render import DepthRenderer, StrandRenderer

    z = np.load(os.path.join(GOLDEN, "gl_raster.npz"))
    H, W = int(z["H"]), int(z["W"])
    cams = [dict(file="v%d" % i, pose=z["cam_pose"][i].tolist(), ndc_prj=z["cam_ndc"][i].tolist())
            for i in range(len(z["cam_pose"]))]
    rec = camera_records(cameras_from_list(cams))
    two = DepthRenderer([(z["v1"], z["f1"]), (z["v2"], z["f2"])], DEV)
    soup = DepthRenderer([(z["soup_v"], z["soup_f"])], DEV)
    strands = StrandRenderer([], z["v1"], z["f1"], DEV)
    strands.line_pts = torch.from_numpy(z["line_pts"]).to(DEV)
    strands.line_tan = torch.from_numpy(z["line_tan"]).to(DEV)
    strands.nseg = len(z["line_pts"]) // 2
    L, ctx = _lib.lib(), _ctx_for(torch.device(DEV))

    def bits(b):
        _lib.check(L.mh_ctx_set_option(ctx, b"raster_subpixel_bits", b))

    try:
        for vi in z["views"]:
            vi = int(vi)
            bits(8)
            check_depth_against_gl(two.render(rec[vi], H, W, pixel_center=0.5).cpu().numpy().reshape(H, W),
                                   z["depth_two_meshes_%d" % vi] * 255, smooth=True)
            check_depth_against_gl(soup.render(rec[vi], H, W, pixel_center=0.5).cpu().numpy().reshape(H, W),
                                   z["depth_soup_%d" % vi] * 255, smooth=False)
            bits(4)
            check_depth_same_grid(two.render(rec[vi], H, W, pixel_center=0.5).cpu().numpy().reshape(H, W),
                                  z["depth_two_meshes_%d" % vi] * 255)
            check_depth_same_grid(soup.render(rec[vi], H, W, pixel_center=0.5).cpu().numpy().reshape(H, W),
                                  z["depth_soup_%d" % vi] * 255)

            def draw(copt, dopt, clear, rule, b):
                bits(b)
                return strands.render(rec[vi], H, W, copt, dopt, clear, pixel_center=0.5, line_width=1,
                                      line_rule=rule).cpu().numpy()

            check_strands_against_gl(draw, z, vi)
    finally:
        bits(8)


def test_hip_rasterisers_against_a_desktop_opengl_with_the_references_own_shaders():
    """tests/golden/gl_mesa.npz: Mesa llvmpipe (OpenGL 4.5 core) compiling the reference's own GLSL, lines at the reference's
    width 3 -- same checks as the C statement's CPU test (tests/gl_checks.py::check_against_desktop_gl)."""
    import os

    from conftest import GOLDEN
    from gl_checks import check_against_desktop_gl
    from monohair_amd.render import DepthRenderer, StrandRenderer

    z = np.load(os.path.join(GOLDEN, "gl_mesa.npz"))
    H, W = int(z["H"]), int(z["W"])
    cams = [dict(file="v%d" % i, pose=z["cam_pose"][i].tolist(), ndc_prj=z["cam_ndc"][i].tolist())
            for i in range(len(z["cam_pose"]))]
    rec = camera_records(cameras_from_list(cams))

    def strand_renderer(with_bust):
        r = StrandRenderer([], z["v1"] if with_bust else np.zeros((0, 3)), z["f1"] if with_bust else np.zeros((0, 3), int), DEV)
        r.line_pts = torch.from_numpy(z["line_pts"]).to(DEV)
        r.line_tan = torch.from_numpy(z["line_tan"]).to(DEV)
        r.nseg = len(z["line_pts"]) // 2
        return r

    rs = {True: strand_renderer(True), False: strand_renderer(False)}
    for vi in z["views"]:
        vi = int(vi)

        def depth_of(v, f):
            return DepthRenderer([(v, f)], DEV).render(rec[vi], H, W, pixel_center=0.5).cpu().numpy().reshape(H, W)

        def draw(width, copt, dopt, clear, with_bust):
            return rs[with_bust].render(rec[vi], H, W, copt, dopt, clear, pixel_center=0.5, line_width=width,
                                        line_rule=0).cpu().numpy()

        check_against_desktop_gl(z, vi, depth_of, draw)
