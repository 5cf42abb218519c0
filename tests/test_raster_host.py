"""CPU: the depth-rasteriser specification (oracle/raster_oracle.c) against analytic answers and against a real OpenGL
implementation (last test).  The reference renders depth with an OpenGL driver (Utils/Render_utils.py:310-347); these
tests pin the written specification: a finely tessellated sphere must reproduce the analytic ray-sphere depth of the
synthetic scene, shared edges must be drawn exactly once, and depth ties go to the earlier primitive."""
import math

import numpy as np
import pytest

import oracle
from monohair_amd import synth
from monohair_amd.camera import camera_records, cameras_from_list


def uv_sphere(radius, n_lat, n_lon):
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
            fs += [(p00, p10, p11), (p00, p11, p01)]
    return np.array(vs, np.float32), np.array(fs, np.int32)


def test_sphere_depth_matches_the_analytic_scene():
    H, W = 240, 136
    cams = synth.make_cameras(24, H, W, scale=1.7)
    rec = camera_records(cameras_from_list(cams))
    v, f = uv_sphere(synth.SPHERE_R, 192, 384)
    for i in (0, 7):
        want = synth.render_view(cams[i], i, H, W, seed=0, quantize=False)[0].numpy()      # analytic, centres at integers
        got, covered = oracle.render_depth(rec[i], v, f, H, W, pixel_center=0.0)
        hit_w, hit_g = want < 255, got < 255
        assert covered == hit_g.sum() and abs(int(hit_g.sum()) - int(hit_w.sum())) <= 0.02 * hit_w.sum()
        both = hit_w & hit_g
        assert (hit_w != hit_g).sum() <= 0.03 * hit_w.sum()          # silhouette pixels only
        assert np.abs(got[both] - want[both]).max() < 0.05 and np.median(np.abs(got[both] - want[both])) < 2e-3
        # OpenGL's sample position (pixel centre +0.5) shifts the image by half a pixel: still the same sphere
        gl, _ = oracle.render_depth(rec[i], v, f, H, W, pixel_center=0.5)
        assert abs(int((gl < 255).sum()) - int(hit_w.sum())) <= 0.03 * hit_w.sum()


def test_watertight_and_tie_rules():
    H, W = 64, 64
    cams = synth.make_cameras(20, H, W, scale=1.0)
    rec = camera_records(cameras_from_list(cams))[0]
    # a fan of triangles around a centre in a plane facing the camera: every covered pixel exactly once
    rng = np.random.default_rng(1)
    n = 17
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    ring = np.stack([0.1 * np.cos(ang), 0.1 * np.sin(ang), np.zeros(n)], 1)
    verts = np.concatenate([[[0.003, -0.002, 0.0]], ring]).astype(np.float32)
    faces = np.array([(0, 1 + k, 1 + (k + 1) % n) for k in range(n)], np.int32)
    whole, cov = oracle.render_depth(rec, verts, faces, H, W)
    per = [oracle.render_depth(rec, verts, faces[k:k + 1], H, W)[1] for k in range(n)]
    assert cov == sum(per) and cov > 100                               # no pixel drawn twice, none lost on shared edges
    # reversed winding and reversed draw order: same coverage (no culling, order-independent fill rule); the values
    # only differ by the rounding of the barycentric sums
    again, _ = oracle.render_depth(rec, verts, faces[::-1, ::-1], H, W)
    assert np.array_equal(whole < 255, again < 255) and np.abs(whole - again).max() < 1e-4
    # two coincident triangles with different colours cannot be told apart by depth -> the first drawn wins; a
    # strictly nearer one replaces
    tri = np.array([[-0.1, -0.1, 0], [0.1, -0.1, 0], [0, 0.1, 0]], np.float32)
    near = tri + np.array([0, 0, 0.05], np.float32) * np.sign(np.array(cams[0]["pose"])[2, 3])
    both = np.concatenate([tri, near]).astype(np.float32)
    a, _ = oracle.render_depth(rec, both, np.array([[0, 1, 2], [3, 4, 5]], np.int32), H, W)
    b, _ = oracle.render_depth(rec, both, np.array([[3, 4, 5], [0, 1, 2]], np.int32), H, W)
    only_near, _ = oracle.render_depth(rec, near, np.array([[0, 1, 2]], np.int32), H, W)
    assert np.array_equal(a, b)
    m = only_near < 255
    assert m.sum() > 50 and np.array_equal(a[m], only_near[m])
    # behind the camera / degenerate / out-of-range indices are skipped, empty input gives the background
    behind = tri + np.array([0, 0, 5.0], np.float32) * np.sign(np.array(cams[0]["pose"])[2, 3])
    e, cov = oracle.render_depth(rec, behind, np.array([[0, 1, 2], [0, 0, 1], [0, 1, 7]], np.int32), H, W)
    assert cov == 0 and (e == 255).all()
    e, cov = oracle.render_depth(rec, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), H, W, channels=3)
    assert e.shape == (H, W, 3) and (e == 255).all()


# ---------------------------------------------------------------------------------------------------------------------
# strand-segment renderer: the written specification on known answers + the reference's buffers
# ---------------------------------------------------------------------------------------------------------------------
def _front_camera(H, W):
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list

    return camera_records(cameras_from_list(synth.make_cameras(4, H, W, scale=0.7, rings=1)))[0]


QUAD = np.array([[-0.1, -0.1, 0], [0.1, -0.1, 0], [0.1, 0.1, 0], [-0.1, 0.1, 0]], np.float32)
QUAD_F = np.array([[0, 1, 2], [0, 2, 3]], np.int32)


def test_strand_vertex_buffers_equal_the_reference():
    """strand_line_buffers == the `Lines` / `tangent` buffers the reference's StrandsObj builds (Render_utils.py:9-29;
    tests/golden/strand_buffers.npz, tools/gen_golden_r2.py strands)."""
    import os

    from conftest import GOLDEN
    from monohair_amd.render import strand_line_buffers

    z = np.load(os.path.join(GOLDEN, "strand_buffers.npz"))
    strands = [z["strand_%02d" % i] for i in range(int(z["n_strands"]))]
    lines, tans = strand_line_buffers(strands)
    assert np.array_equal(lines, z["Lines"].astype(np.float32))        # the reference uploads them as 'f4' (:33-34)
    assert np.array_equal(tans, z["tangent"].astype(np.float32))


def _expected_2theta(direction):
    """colour option 2 of a world direction (dx, dy, 0) seen by the front camera: Tangent_2d lives in NDC, where the
    camera's fx, fy scale the two axes (Render_utils.py:62-66), and NDC y points up while world y does too"""
    from monohair_amd import synth

    fx, fy = synth.make_cameras(4, 160, 120, scale=0.7, rings=1)[0]["ndc_prj"][:2]
    tx, ty = fx * direction[0], fy * direction[1]
    s2 = tx * tx + ty * ty
    return ((tx * tx - ty * ty) / s2 + 1) / 2, (2 * tx * ty / s2 + 1) / 2


@pytest.mark.parametrize("direction", [(1, 0, 0), (0, 1, 0), (1, 1, 0), (1, -1, 0)])
def test_strand_colours_widths_and_depth_test(direction):
    expect = _expected_2theta(direction)
    """A straight strand in front of a quad: width-3 band in the minor direction, colour option 2 = ((cos 2th, sin 2th, 0)
    + (1,1,0))/2 of its image direction (the same for both senses), option 3 white, option 0 depth/2; behind the quad it
    is hidden; bust options 0/1/2 = depth/2, black, white; background = clear colour."""
    from monohair_amd.render import strand_line_buffers

    H, W = 160, 120
    rec = _front_camera(H, W)
    d = np.array(direction, np.float64) / np.linalg.norm(direction)
    strand = np.linspace(-0.06, 0.06, 25)[:, None] * d[None] + np.array([0, 0, 0.03])
    for pts in (strand, strand[::-1]):
        lp, lt = strand_line_buffers([pts])
        rgb, prim, owned = oracle.render_strands(rec, QUAD, QUAD_F, lp, lt, H, W, 0.5, 3, 2, 1, 0.0)
        on = prim >= len(QUAD_F)
        assert owned == on.sum() and owned >= 24
        # image x runs with world x, image rows run against world y in this camera; 2*theta colouring is sense-free
        assert np.allclose(rgb[on][:, 0], expect[0], atol=0.02) and np.allclose(rgb[on][:, 2], 0.0)
        assert np.allclose(rgb[on][:, 1], expect[1], atol=0.02), (rgb[on][:, 1].min(), rgb[on][:, 1].max())
        assert np.all(rgb[(prim >= 0) & ~on] == 0.0) and np.all(rgb[prim < 0] == 0.0)
    # width: every fragment column of an axis-aligned strand carries exactly 3 pixels
    if direction in ((1, 0, 0), (0, 1, 0)):
        cols = on.sum(0 if direction[0] else 1)
        assert set(cols[cols > 0].tolist()) == {3}
    white, _, _ = oracle.render_strands(rec, QUAD, QUAD_F, lp, lt, H, W, 0.5, 3, 3, 1, 0.0)
    assert np.all(white[on] == 1.0)
    depth, _, _ = oracle.render_strands(rec, QUAD, QUAD_F, lp, lt, H, W, 0.5, 3, 0, 2, 1.0)
    assert np.allclose(depth[on][:, 0], (0.8 - 0.03) / 2.0, atol=2e-3)      # camera ring radius 0.8 m, strand 3 cm nearer
    assert np.all(depth[(prim >= 0) & ~on] == 1.0) and np.all(depth[prim < 0] == 1.0)
    bust, pb, _ = oracle.render_strands(rec, QUAD, QUAD_F, lp, lt, H, W, 0.5, 3, -1, 0, 1.0)     # strands not drawn
    assert (pb >= len(QUAD_F)).sum() == 0 and np.allclose(bust[pb >= 0][:, 0], 0.8 / 2.0, atol=2e-3)
    # behind the quad: hidden by the depth test
    lp2, lt2 = strand_line_buffers([strand - np.array([0, 0, 0.06])])
    _, p2, owned2 = oracle.render_strands(rec, QUAD, QUAD_F, lp2, lt2, H, W, 0.5, 3, 3, 1, 0.0)
    assert owned2 == 0 and (p2 >= 0).sum() == (prim >= 0).sum() - 0 * owned


def test_oracle_rasterisers_against_real_opengl():
    """The C statements of both rasterisers against images drawn by a real OpenGL implementation (Google SwiftShader,
    tests/golden/gl_raster.npz, tools/gen_golden_gl.py): the reference's triangle pass with two intersecting meshes and
    a random triangle soup, and its line pass over the bust in the three colourings (see tests/gl_checks.py for what is
    compared and why the comparison has tolerances).  pixel_center 0.5 is GL's sample position: 0.0 does not fit."""
    import os

    from conftest import GOLDEN
    from gl_checks import check_depth_against_gl, check_depth_same_grid, check_strands_against_gl

    z = np.load(os.path.join(GOLDEN, "gl_raster.npz"))
    H, W = int(z["H"]), int(z["W"])
    cams = [dict(file="v%d" % i, pose=z["cam_pose"][i].tolist(), ndc_prj=z["cam_ndc"][i].tolist())
            for i in range(len(z["cam_pose"]))]
    rec = camera_records(cameras_from_list(cams))
    v = np.concatenate([z["v1"], z["v2"]])
    f = np.concatenate([z["f1"], z["f2"] + len(z["v1"])])
    try:
        for vi in z["views"]:
            vi = int(vi)
            oracle.set_subpixel_bits(8)
            got, _ = oracle.render_depth(rec[vi], v, f, H, W, pixel_center=0.5)
            check_depth_against_gl(got.reshape(H, W), z["depth_two_meshes_%d" % vi] * 255, smooth=True)
            shifted, _ = oracle.render_depth(rec[vi], v, f, H, W, pixel_center=0.0)
            assert ((shifted.reshape(H, W) < 255) != (z["depth_two_meshes_%d" % vi] < 1.0)).sum() > 100
            got, _ = oracle.render_depth(rec[vi], z["soup_v"], z["soup_f"], H, W, pixel_center=0.5)
            check_depth_against_gl(got.reshape(H, W), z["depth_soup_%d" % vi] * 255, smooth=False)
            oracle.set_subpixel_bits(4)
            got, _ = oracle.render_depth(rec[vi], v, f, H, W, pixel_center=0.5)
            check_depth_same_grid(got.reshape(H, W), z["depth_two_meshes_%d" % vi] * 255)
            got, _ = oracle.render_depth(rec[vi], z["soup_v"], z["soup_f"], H, W, pixel_center=0.5)
            check_depth_same_grid(got.reshape(H, W), z["depth_soup_%d" % vi] * 255)

            def draw(copt, dopt, clear, rule, bits):
                oracle.set_subpixel_bits(bits)
                return oracle.render_strands(rec[vi], z["v1"], z["f1"], z["line_pts"], z["line_tan"], H, W, 0.5, 1, copt,
                                             dopt, clear, line_rule=rule)[0]

            check_strands_against_gl(draw, z, vi)
    finally:
        oracle.set_subpixel_bits(8)


def test_oracle_rasterisers_against_a_desktop_opengl_with_the_references_own_shaders():
    """tests/golden/gl_mesa.npz: Mesa llvmpipe (OpenGL 4.5 core) compiling the reference's own GLSL, lines at the reference's
    width 3 -- the wide-line rule, which SwiftShader could not pin (tests/gl_checks.py::check_against_desktop_gl)."""
    import os

    from conftest import GOLDEN
    from gl_checks import check_against_desktop_gl

    z = np.load(os.path.join(GOLDEN, "gl_mesa.npz"))
    H, W = int(z["H"]), int(z["W"])
    cams = [dict(file="v%d" % i, pose=z["cam_pose"][i].tolist(), ndc_prj=z["cam_ndc"][i].tolist())
            for i in range(len(z["cam_pose"]))]
    rec = camera_records(cameras_from_list(cams))
    none_v, none_f = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    oracle.set_subpixel_bits(8)
    for vi in z["views"]:
        vi = int(vi)

        def depth_of(v, f):
            return oracle.render_depth(rec[vi], v, f, H, W, pixel_center=0.5)[0].reshape(H, W)

        def draw(width, copt, dopt, clear, with_bust):
            bv, bf = (z["v1"], z["f1"]) if with_bust else (none_v, none_f)
            return oracle.render_strands(rec[vi], bv, bf, z["line_pts"], z["line_tan"], H, W, 0.5, width, copt, dopt, clear,
                                         line_rule=0)[0]

        check_against_desktop_gl(z, vi, depth_of, draw)


def test_line_rule_known_answers():
    """The fragments of single horizontal segments, start x0 = 5 + f0, end x1 = 12 + f1 (GL window coordinates, pixel
    centres at +0.5).  line_rule 1 = every diamond |dx| + |dy| < 1/2 the closed segment touches -- the columns below are
    what Google SwiftShader drew for these segments (probed with tools/gl_ref; exact diamond-boundary cases left out);
    line_rule 0 = OpenGL 4.6 14.5.1: the same minus the pixel whose diamond contains the END point."""
    H = W = 32
    cam = synth.make_cameras(20, H, W, scale=1.0)[0]
    rec = camera_records(cameras_from_list([cam]))[0]
    c2w = np.array(cam["pose"], np.float64)
    fx, fy, cx, cy = cam["ndc_prj"]

    def world(col, row, depth=1.0):          # the 3D point that projects to pixel coordinates (col, row)
        z = -depth
        u, v = 1.0 - 2.0 * col / W, 2.0 * row / H - 1.0
        x, y = (u - cx) * z / fx, (v - cy) * z / fy
        return (c2w @ np.array([x, y, z, 1.0]))[:3]

    def columns(x0, x1, y_gl, rule):
        row = H - y_gl                        # image rows run downwards, GL's y upwards
        p = np.stack([world(x0, row), world(x1, row)]).astype(np.float32)
        t = np.stack([p[1] - p[0]] * 2).astype(np.float32)
        img, prim, _ = oracle.render_strands(rec, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), p, t, H, W,
                                             0.5, 1, 3, 1, 0.0, line_rule=rule)
        rows, cols = np.nonzero(prim >= 0)
        assert len(set(rows.tolist())) <= 1 and (not len(rows) or rows[0] == int(np.floor(row)))
        return (int(cols.min()), int(cols.max())) if len(cols) else None

    # through the pixel centres (y = 10.5): the diamonds reach the pixel borders on this line
    for f0, first in ((0.1, 5), (0.25, 5), (0.45, 5), (0.55, 5), (0.7, 5), (0.9, 5)):
        for f1, last_touch, last_exit in ((0.1, 12, 11), (0.25, 12, 11), (0.45, 12, 11), (0.55, 12, 11), (0.9, 12, 11)):
            assert columns(5 + f0, 12 + f1, 10.5, 1) == (first, last_touch), (f0, f1)
            assert columns(5 + f0, 12 + f1, 10.5, 0) == (first, last_exit), (f0, f1)
    # 0.2 below the centres (y = 10.3): the diamond of pixel i is cut at i + 0.2 .. i + 0.8
    for f0, first in ((0.1, 5), (0.25, 5), (0.45, 5), (0.55, 5), (0.7, 5), (0.9, 6)):
        for f1, last_touch, last_exit in ((0.1, 11, 11), (0.25, 12, 11), (0.45, 12, 11), (0.55, 12, 11), (0.7, 12, 11),
                                          (0.9, 12, 12)):
            assert columns(5 + f0, 12 + f1, 10.3, 1) == (first, last_touch), (f0, f1)
            assert columns(5 + f0, 12 + f1, 10.3, 0) == (first, last_exit), (f0, f1)
    # drawn backwards, the roles of the ends swap: the start pixel stays, the end pixel (now on the left) goes
    assert columns(12.45, 5.45, 10.5, 1) == (5, 12) and columns(12.45, 5.45, 10.5, 0) == (6, 12)
    # a segment inside one diamond: touched, but not left
    assert columns(8.4, 8.6, 10.5, 1) == (8, 8) and columns(8.4, 8.6, 10.5, 0) is None
