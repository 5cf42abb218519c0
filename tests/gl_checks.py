"""What "equal to a real OpenGL implementation" means for the two rasterisers (shared by the CPU test of the oracle and the
GPU test of the HIP kernels; fixtures: tests/golden/gl_raster.npz from Google SwiftShader, tools/gen_golden_gl.py).
OpenGL fixes the sample positions, the one-fragment-per-shared-edge-pixel rule, perspective-correct varyings, the LESS
depth test in draw order and the diamond rule of lines; it leaves to the implementation the sub-pixel grid the vertices
are snapped to (SwiftShader: 1/16 pixel, ours: 1/256) and the rounding of the interpolation.  So: coverage may differ on
a few silhouette / line-edge pixels, values by the attribute gradient times a fraction of a pixel."""
import numpy as np


def check_depth_same_grid(mine, gl):
    """With window positions snapped to SwiftShader's own sub-pixel grid (1/16 pixel) the triangle pass is the same image:
    identical coverage -- silhouettes, shared edges, intersection lines of the depth test -- and values to 4e-3 of 255
    (the rounding of two different interpolation formulas)."""
    assert np.array_equal(mine < 255, gl < 255)
    assert np.abs(mine - gl).max() < 4e-3


def check_depth_against_gl(mine, gl, smooth):
    """The shipped 1/256-pixel grid.  mine / gl: [H,W] depth images in the reference's units (value = -z_camera / 2 * 255,
    background 255)."""
    hm, hg = mine < 255, gl < 255
    n = int(hg.sum())
    assert n > 5000
    assert int((hm != hg).sum()) <= 0.002 * n + 4                       # silhouette pixels only (measured: <= 14 of 9500)
    both = hm & hg
    d = np.abs(mine[both] - gl[both])
    assert np.median(d) < 0.004                                         # 1.5e-5 of the value range
    if smooth:                                                          # one closed surface in front: no z-fighting
        assert d.max() < 0.5 and np.mean(d > 0.05) < 0.02
    else:                                                               # intersecting random triangles: the depth test
        assert np.mean(d > 0.05) < 0.02                                 # may pick the other one on an intersection line


def check_strands_against_gl(draw, z, vi):
    """draw(color_option, depth_option, clear, line_rule, subpixel_bits) -> [H,W,3] image of 1-pixel lines over the bust;
    z: the fixture; vi: view index."""
    gm = z["strand_mask_w1_%d" % vi] > 0.5
    n = int(gm.sum())
    assert n > 500
    # SwiftShader's grid (4 sub-pixel bits) and its end-pixel rule: the same line pixels, all of them
    same = draw(3, 1, 0.0, 1, 4)[..., 0] > 0.5
    assert np.array_equal(same, gm)
    # colours and depth: the GLSL shader's atan / cos / sin against the algebraic form, float rounding only
    col = draw(2, 1, 0.0, 1, 4)
    d = np.abs(col[gm] - z["strand_color_w1_%d" % vi][gm]).max(1)
    assert np.median(d) < 1e-6 and d.max() < 1e-3                      # measured: 1e-7 / 2.4e-4
    dep = draw(0, 2, 1.0, 1, 4)[..., 0]
    dd = np.abs(dep[gm] - z["strand_depth_w1_%d" % vi][gm])
    assert np.median(dd) < 1e-7 and dd.max() < 1e-6                    # measured: 3e-8 / 1.8e-7
    assert np.array_equal(dep == 1.0, z["strand_depth_w1_%d" % vi] == 1.0)     # background and white bust pixels
    # OpenGL's own rule (line_rule 0) draws a subset: it only drops pixels that hold the end point of a segment
    exit4 = draw(3, 1, 0.0, 0, 4)[..., 0] > 0.5
    assert not (exit4 & ~same).any() and 0 < int((same & ~exit4).sum()) <= 0.04 * n
    # the shipped grid (8 bits): the same picture up to the snapping
    touch = draw(3, 1, 0.0, 1, 8)[..., 0] > 0.5
    assert int((touch != gm).sum()) <= 0.04 * n and abs(int(touch.sum()) - n) <= 0.01 * n
