"""CPU: property-based checks of the host-side formats and helpers (hypothesis): whatever goes in comes back out."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st
from hypothesis.extra import numpy as hnp

from monohair_amd import mapspack
from monohair_amd import pmvo_utils as U

SET = dict(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])


@settings(**SET)
@given(v=st.integers(1, 4), h=st.integers(1, 9), w=st.integers(1, 9), seed=st.integers(0, 2 ** 31 - 1))
def test_maps_pack_is_lossless(tmp_path, v, h, w, seed):
    rng = np.random.default_rng(seed)
    views = ["view_%d" % i for i in range(v)]
    o, c, m = (rng.integers(0, 256, (v, h, w)).astype(np.uint8) for _ in range(3))
    d = rng.normal(size=(v, h, w)).astype(np.float32)
    p = str(tmp_path / ("p%d.mhpk" % seed))
    mapspack.write_pack(p, views, o, c, m, d)
    got = mapspack.read_pack(p)
    assert got["views"] == views
    for name, want in (("ori", o), ("conf", c), ("mask", m), ("depth", d)):
        assert np.array_equal(np.asarray(got[name]), want)
    order = list(reversed(views))
    sub = mapspack.read_pack(p, views=order)
    assert all(np.array_equal(np.asarray(sub["depth"][i]), d[v - 1 - i]) for i in range(v))


@settings(**SET)
@given(n=st.integers(0, 60), g=st.tuples(st.integers(1, 12), st.integers(1, 12), st.integers(1, 12)), seed=st.integers(0, 10 ** 6))
def test_sparse_mat_files_equal_dense_savemat(tmp_path, n, g, seed):
    import scipy.io

    rng = np.random.default_rng(seed)
    vox = np.stack([rng.integers(0, g[i], n) for i in range(3)], 1).reshape(n, 3)
    ori = rng.normal(size=(n, 3)).astype(np.float32)
    a = tmp_path / ("a%d" % seed)
    b = tmp_path / ("b%d" % seed)
    a.mkdir(exist_ok=True), b.mkdir(exist_ok=True)
    U.save_ori_occ_mat_sparse(str(a), g, vox, ori)
    occ, dense = U.dense_from_sparse(g, vox, ori)
    U.save_ori_occ_mat(str(b), occ, dense)
    for f, k in (("Ori3D.mat", "Ori"), ("Occ3D.mat", "Occ")):
        assert np.array_equal(scipy.io.loadmat(a / f)[k], scipy.io.loadmat(b / f)[k])


@settings(**SET)
@given(pts=hnp.arrays(np.float64, st.tuples(st.integers(1, 40), st.just(3)),
                      elements=st.floats(-0.3, 0.3, allow_nan=False, width=64)))
def test_p2v_matches_its_definition(pts):
    g = np.array([256, 256, 192])
    vmin, vs = np.array([-0.32, -0.32, -0.24]), 0.0025
    x, y, z = U.p2v(pts.copy(), vmin, vs, g)
    q = pts.copy()
    q[:, 1:] *= -1
    want = np.clip(np.round((q - vmin) / vs).astype(np.int32), 0, g - 1)
    assert np.array_equal(np.stack([x, y, z], 1), want)


@settings(**SET)
@given(segs=st.lists(st.integers(1, 30), min_size=1, max_size=12), seed=st.integers(0, 10 ** 6))
def test_hair_file_roundtrip(tmp_path, segs, seed):
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(sum(segs), 3)).astype(np.float32)
    p = str(tmp_path / ("s%d.hair" % seed))
    U.write_strand(pts, p, segs)
    got_segs, got_pts = U.load_strand(p)
    assert list(got_segs) == list(segs) and np.array_equal(np.asarray(got_pts, np.float32).reshape(-1, 3), pts)


@settings(**SET)
@given(nv=st.integers(3, 40), nf=st.integers(0, 60), seed=st.integers(0, 10 ** 6), slashes=st.booleans())
def test_obj_reader_reads_what_was_written(tmp_path, nv, nf, seed, slashes):
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(nv, 3))
    f = rng.integers(0, nv, (nf, 3))
    p = tmp_path / ("m%d.obj" % seed)
    face = "f %d/1 %d/2 %d/3\n" if slashes else "f %d %d %d\n"
    p.write_text("".join("v %.17g %.17g %.17g\n" % tuple(x) for x in v) + "".join(face % tuple(t + 1) for t in f))
    gv, gf = U.read_obj(str(p))
    assert np.array_equal(gv, v) and np.array_equal(gf.reshape(-1, 3), f)
