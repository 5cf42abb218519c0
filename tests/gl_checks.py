"""What "equal to a real OpenGL implementation" means for the two rasterisers (shared by the CPU test of the oracle and the
GPU test of the HIP kernels; fixtures: tests/golden/gl_raster.npz from Google SwiftShader, tools/gen_golden_gl.py).
OpenGL fixes the sample positions, the one-fragment-per-shared-edge-pixel rule, perspective-correct varyings, the LESS
depth test in draw order and the diamond rule of lines; it leaves to the implementation the sub-pixel grid the vertices
are snapped to (SwiftShader: 1/16 pixel, ours: 1/256) and the rounding of the interpolation.  So: coverage may differ on
a few silhouette / line-edge pixels, values by the attribute gradient times a fraction of a pixel."""
import numpy as np


def check_depth_against_gl(mine, gl, smooth):
    """mine / gl: [H,W] depth images in the reference's units (value = -z_camera / 2 * 255, background 255)."""
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
    """draw(color_option, depth_option, clear, line_rule) -> [H,W,3] image of 1-pixel lines over the bust;
    z: the fixture; vi: view index."""
    gm = z["strand_mask_w1_%d" % vi] > 0.5
    n = int(gm.sum())
    assert n > 500
    touch = draw(3, 1, 0.0, 1)[..., 0] > 0.5                            # every touched diamond: SwiftShader's rule
    exitr = draw(3, 1, 0.0, 0)[..., 0] > 0.5                            # OpenGL's diamond-exit rule (shipped)
    assert int((touch != gm).sum()) <= 0.04 * n                         # measured 2.5-3.5 %: sub-pixel snapping
    assert abs(int(touch.sum()) - n) <= 0.01 * n                        # no systematic surplus or deficit
    # the specified rule draws a subset: it only drops pixels that hold the end point of a segment
    assert not (exitr & ~touch).any() and 0 < int((touch & ~exitr).sum()) <= 0.04 * n
    both = touch & gm
    col = draw(2, 1, 0.0, 1)
    d = np.abs(col[both] - z["strand_color_w1_%d" % vi][both]).max(1)
    assert np.median(d) < 3e-3 and np.mean(d > 0.02) < 0.06             # undirected-orientation colours (measured 1e-3 / 0.04)
    dep = draw(0, 2, 1.0, 1)[..., 0]
    dd = np.abs(dep[both] - z["strand_depth_w1_%d" % vi][both])
    assert np.median(dd) < 1e-3 and np.mean(dd > 0.01) < 0.05           # depth / 2 along the strands
    # pixels of the bust that no strand covers are black / white exactly as in GL
    bust_gl = (z["strand_depth_w1_%d" % vi] == 1.0) & ~gm
    assert np.mean(dep[bust_gl & ~touch] == 1.0) > 0.999
