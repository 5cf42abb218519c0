"""CPU-only tests of the host logic: options grammar, file formats, loaders, distributed plumbing (gloo x2)."""
import os
import struct
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_options_grammar_and_parent_inheritance(tmp_path, monkeypatch):
    from monohair_amd import options

    (tmp_path / "base.yaml").write_text("seed: 0\ncpu:\ngpu: 0\nname: run\nPMVO:\n  patch_size: 9\n  optimize: true\n"
                                        "  infer_inner: true\ndata:\n  root: data\n  case:\n")
    (tmp_path / "case.yaml").write_text("_parent_: %s\ndata:\n  case: c1\nPMVO:\n  patch_size: 7\n" %
                                        (tmp_path / "base.yaml"))
    cmd = options.parse_arguments(["--yaml=%s" % (tmp_path / "case"), "--PMVO.infer_inner!", "--PMVO.optimize=",
                                   "--data.root=elsewhere", "--brand.new=1"])
    assert cmd.PMVO.infer_inner is False and cmd.PMVO.optimize is None
    opt = options.set(cmd)      # unknown key "brand.new" must not block (non-interactive -> auto "y")
    assert opt.PMVO.patch_size == 7 and opt.data.case == "c1" and opt.data.root == "elsewhere"
    assert opt.PMVO.optimize is None and opt.brand.new == 1 and opt.name == "run"
    assert opt.device in ("cpu",) or opt.device.startswith("cuda:")
    opt.output_path = str(tmp_path)
    options.save_options_file(opt)
    opt.PMVO.patch_size = 5
    options.save_options_file(opt)   # differing file: must not block either
    import yaml

    assert yaml.safe_load(open(tmp_path / "options.yaml"))["PMVO"]["patch_size"] == 5
    cmd2 = options.parse_arguments(["--yaml=%s" % (tmp_path / "case"), "--seed=3"])
    assert options.set(cmd2).name == "run_seed3"       # options.py:99-105


def test_repo_configs_load():
    from monohair_amd import options

    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        opt = options.set(options.parse_arguments(["--yaml=configs/reconstruct/big_wavy1"]))
    finally:
        os.chdir(cwd)
    assert opt.PMVO.patch_size == 7 and opt.PMVO.threshold == 0.025 and opt.PMVO.conf_threshold == 0.15
    assert opt.data.image_size == [1920, 1080] and opt.data.depth_path == "render_depth"


def test_hair_file_layout_roundtrip(tmp_path):
    from monohair_amd.pmvo_utils import load_strand, save_hair_strands

    rng = np.random.default_rng(0)
    strands = [rng.normal(size=(n, 3)).astype(np.float32) for n in (5, 1, 17)]
    p = str(tmp_path / "x.hair")
    save_hair_strands(p, strands, np.zeros(3), translate=False)
    raw = open(p, "rb").read()
    assert struct.unpack("<II", raw[:8]) == (3, 23)
    assert struct.unpack("<HHH", raw[8:14]) == (5, 1, 17)
    assert len(raw) == 8 + 6 + 23 * 12
    seg, pts = load_strand(p)
    assert seg == [5, 1, 17] and np.allclose(pts, np.concatenate(strands))


def test_mat_layout_roundtrip(tmp_path):
    from monohair_amd.pmvo_utils import get_ground_truth_3D_occ, get_ground_truth_3D_ori, save_ori_occ_mat

    g = (8, 6, 4)
    rng = np.random.default_rng(1)
    occ = (rng.random(g) > 0.5).astype(np.float64)
    ori = rng.normal(size=g + (3,))
    save_ori_occ_mat(str(tmp_path), occ, ori)
    o = get_ground_truth_3D_ori(str(tmp_path / "Ori3D.mat"))     # [Z,Y,X,3]
    c = get_ground_truth_3D_occ(str(tmp_path / "Occ3D.mat"))     # [Z,Y,X,1]
    assert o.shape == (4, 6, 8, 3) and c.shape == (4, 6, 8, 1)
    assert np.allclose(o, ori.transpose(2, 1, 0, 3).astype(np.float32))
    assert np.allclose(c[..., 0], occ.transpose(2, 1, 0).astype(np.float32))


def test_loaders_formulas(tmp_path):
    from PIL import Image

    from monohair_amd.pmvo_utils import Load_Ori_And_Conf, load_depth, load_mask

    for d in ("best_ori", "conf", "hair_mask", "render_depth"):
        os.makedirs(tmp_path / d)
    deg = np.array([[0, 45], [90, 179]], np.uint8)
    Image.fromarray(deg).save(tmp_path / "best_ori" / "v0.png")
    Image.fromarray(np.array([[0, 51], [102, 255]], np.uint8)).save(tmp_path / "conf" / "v0.png")
    m = np.zeros((2, 2, 3), np.uint8)
    m[0, 0] = 49
    m[0, 1] = 50
    m[1, 1] = 255
    Image.fromarray(m).save(tmp_path / "hair_mask" / "v0.png")
    np.save(tmp_path / "render_depth" / "v0.npy", np.full((2, 2, 3), 255.0))
    cam = {"v0": None}
    Ori, Conf = Load_Ori_And_Conf(cam, str(tmp_path / "best_ori"), str(tmp_path / "conf"))
    th = (180 - deg.astype(np.float64)) / 180 * np.pi
    assert np.array_equal(Ori["v0"], np.stack([np.sin(th), np.cos(th)], -1))
    assert np.array_equal(Conf["v0"], np.array([[0, 51], [102, 255]]) / 255.0)
    mask = load_mask(cam, str(tmp_path / "hair_mask"))["v0"]
    assert mask[0, 0, 0] == 0 and mask[0, 1, 0] == 50 / 255.0 and mask[1, 1, 0] == 1.0
    assert load_depth(cam, str(tmp_path / "render_depth"))["v0"].dtype == np.float32


def test_obj_reader_and_sampling(tmp_path):
    from monohair_amd.pmvo_utils import SamplePointsAroundmesh, load_bust, sample_points_uniformly

    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1/1 3/2 4/3\n")
    v, f, n = load_bust(str(tmp_path / "m.obj"))
    assert v.shape == (4, 3) and f.tolist() == [[0, 1, 2], [0, 2, 3]] and n.shape == (4, 3)
    s = sample_points_uniformly(v, f, 1000)
    assert s.shape == (1000, 3) and s.min() >= 0 and s.max() <= 1
    np.random.seed(0)
    pts = SamplePointsAroundmesh(np.array([[0.0, 0.0, 0.0], [0.01, 0.0, 0.0]]), np.array([-0.32, -0.32, -0.24]),
                                 0.00125, num_per_grid=4, grid_resolution=[512, 512, 384])
    assert pts.shape == (8, 3)


def test_obj_reader_bulk_path_equals_line_loop(tmp_path):
    """read_obj parses regular files in bulk and falls back to the line loop for polygons / relative indices /
    ragged vertex lines: same arrays either way, and 17-digit coordinates survive exactly."""
    from monohair_amd.pmvo_utils import _read_obj_slow, read_obj

    rng = np.random.default_rng(0)
    v = rng.normal(size=(300, 3))
    f = rng.integers(0, 300, (500, 3))
    variants = {
        "plain": ("v %.17g %.17g %.17g\n", "f %d %d %d\n"),
        "crlf": ("v %.17g %.17g %.17g\r\n", "f %d %d %d\r\n"),
        "colour": ("v %.17g %.17g %.17g 0.5 0.25 1\n", "f %d %d %d\n"),
        "slashes": ("v %.17g %.17g %.17g\n", "f %d/1/1 %d//2 %d/3\n"),
    }
    for name, (vf, ff) in variants.items():
        p = tmp_path / (name + ".obj")
        p.write_text("# c\nmtllib m.mtl\n" + "".join(vf % tuple(x) for x in v) + "vn 0 0 1\n" +
                     "".join(ff % tuple(t + 1) for t in f))
        got_v, got_f = read_obj(str(p))
        assert np.array_equal(got_v, v) and np.array_equal(got_f, f), name
    quad = tmp_path / "quad.obj"
    quad.write_text("".join("v %.17g %.17g %.17g\n" % tuple(x) for x in v) + "f 1 2 3 4\nf -1 -2 -3\n")
    got_v, got_f = read_obj(str(quad))
    ref_v, ref_f = _read_obj_slow(quad.read_bytes().split(b"\n"))
    assert np.array_equal(got_v, ref_v) and got_f.tolist() == ref_f.tolist() == [[0, 1, 2], [0, 2, 3], [299, 298, 297]]


_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
from monohair_amd import dist as mdist
dist.init_process_group(backend="gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
r = dist.get_rank()
# --- map_chunks: results come back in chunk order on every rank, whichever rank produced them
chunks = [np.arange(n * 3, dtype=np.float64).reshape(n, 3) + 100 * i for i, n in enumerate([5, 0, 7, 3, 4])]
out = mdist.map_chunks(chunks, lambda c: torch.from_numpy(c).float() * 2, "cpu", lambda: torch.empty((0, 3)))
for c, o in zip(chunks, out):
    assert torch.equal(o, torch.from_numpy(c).float() * 2)
# --- all_gather_views: every rank ends up with all per-view planes, in view order
V = 7
local = [torch.full((2, 3, 4), i, dtype=torch.uint8) for i in range(V) if mdist.owner(i) == r]
allv = mdist.all_gather_views(local, V, (2, 3, 4), torch.uint8, "cpu")
assert allv.shape == (V, 2, 3, 4) and all(int(allv[i].max()) == i and int(allv[i].min()) == i for i in range(V))
# --- all_gather_rows_inplace (refine's per-chunk exchange): rows each rank wrote end up on every rank; the in-place
# capability probe is decided once and agreed by all ranks (gloo: the staged path is used, the probe must still agree)
buf = torch.zeros((20 + 2 * 4, 3))
buf[:] = torch.arange(28, dtype=torch.float32)[:, None]
buf[8 + r * 4: 8 + (r + 1) * 4] = 1000.0 + r                     # chunk rows [8, 16): 4 rows per rank
mdist.all_gather_rows_inplace(buf, 8, 4)
assert torch.equal(buf[8:12], torch.full((4, 3), 1000.0)) and torch.equal(buf[12:16], torch.full((4, 3), 1001.0))
assert torch.equal(buf[:8, 0], torch.arange(8.0)) and torch.equal(buf[16:, 0], torch.arange(16.0, 28.0))
try:
    flag = mdist.inplace_gather_supported("cpu")
except Exception as e:          # a backend without all_reduce on this tensor would be a bug of the probe
    raise AssertionError("probe raised: %%r" %% (e,))
flags = [None, None]
dist.all_gather_object(flags, flag)
assert flags[0] == flags[1], flags
assert mdist.inplace_gather_supported("cpu") == flag                # cached
# --- voxel_fit_reduced: disjoint ownership + ONE reduce == single-process fit, bit for bit
import oracle
def fit(p, o, device, vmin, vsize, g, dense=True):
    occ, ori = oracle.voxel_fit(p, o, vmin, vsize, g)
    vx = np.argwhere(occ != 0)
    return dict(voxels=torch.from_numpy(vx), ori=torch.from_numpy(ori[vx[:, 0], vx[:, 1], vx[:, 2]].astype(np.float32)),
                occ=occ, ori_dense=ori)
rng = np.random.default_rng(5)
pts = rng.uniform(-0.1, 0.1, size=(4000, 3)); ori = rng.normal(size=(4000, 3)).astype(np.float32)
g = [64, 64, 48]
occ, vol = mdist.voxel_fit_reduced(pts, ori, "cpu", [-0.32, -0.32, -0.24], 0.01, g, fit=fit)
if r == 0:
    occ1, vol1 = oracle.voxel_fit(pts.copy(), ori.copy(), [-0.32, -0.32, -0.24], 0.01, g)
    assert np.array_equal(occ, occ1), "occupancy differs from the single-process fit"
    assert np.array_equal(vol.astype(np.float32), vol1.astype(np.float32)), "volume differs"
vx, vo = mdist.voxel_fit_reduced(pts, ori, "cpu", [-0.32, -0.32, -0.24], 0.01, g, fit=fit, sparse=True)
if r == 0:
    from monohair_amd.pmvo_utils import dense_from_sparse
    occ2, vol2 = dense_from_sparse(g, vx, vo)
    assert np.array_equal(occ2, occ1) and np.array_equal(vol2.astype(np.float32), vol1.astype(np.float32)), "sparse"
else:
    assert len(vx) == 0
# the dense-reduce comparison mode (a full volume on every rank, summed into rank 0) gives the same volume
os.environ["MH_VOLUME_EXCHANGE"] = "dense"
occ3, vol3 = mdist.voxel_fit_reduced(pts, ori, "cpu", [-0.32, -0.32, -0.24], 0.01, g, fit=fit)
os.environ["MH_VOLUME_EXCHANGE"] = "torch"
if r == 0:
    assert np.array_equal(occ3, occ1) and np.array_equal(vol3.astype(np.float32), vol1.astype(np.float32)), "dense mode"
    print("DIST_OK", int(occ.sum()))
dist.barrier()
dist.destroy_process_group()
'''


def test_distributed_sharding_and_volume_reduce_gloo(tmp_path):
    """world_size 2 on CPU (gloo): chunk dealing + all_gather ordering, and the single volume reduce."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]


def test_pixel_code_lut_matches_the_float_loaders(tmp_path):
    """map_code_lut()[code] must be exactly what Load_Ori_And_Conf / load_mask + the float32 cast of the constructor
    give for a pixel of that code (Utils/PMVO_utils.py:255-313, PMVO.py:23-26); load_maps_u8 returns the raw codes."""
    from PIL import Image

    from monohair_amd import pmvo_utils as U

    codes = np.arange(256, dtype=np.uint8).reshape(16, 16)
    for d in ("best_ori", "conf", "hair_mask"):
        os.makedirs(tmp_path / d)
    Image.fromarray(codes).save(tmp_path / "best_ori" / "v0.png")
    Image.fromarray(codes.T.copy()).save(tmp_path / "conf" / "v0.png")
    bgr = np.stack([codes[::-1], codes, 255 - codes], -1)          # channel 0 (B) is the one PMVO reads
    Image.fromarray(bgr[..., ::-1].copy()).save(tmp_path / "hair_mask" / "v0.png")
    cam = {"v0": None}
    Ori, Conf = U.Load_Ori_And_Conf(cam, str(tmp_path / "best_ori"), str(tmp_path / "conf"))
    mask = U.load_mask(cam, str(tmp_path / "hair_mask"))
    o8, c8, m8 = U.load_maps_u8(cam, str(tmp_path / "best_ori"), str(tmp_path / "conf"), str(tmp_path / "hair_mask"),
                                threads=2)
    assert np.array_equal(o8["v0"], codes) and np.array_equal(c8["v0"], codes.T) and np.array_equal(m8["v0"], codes[::-1])
    lut = U.map_code_lut()
    assert lut.shape == (256, 4) and lut.dtype == np.float32
    assert np.array_equal(lut[o8["v0"]][..., :2], Ori["v0"].astype(np.float32))
    assert np.array_equal(lut[c8["v0"]][..., 2], Conf["v0"].astype(np.float32))
    assert np.array_equal(lut[m8["v0"]][..., 3], mask["v0"][..., 0].astype(np.float32))
    assert lut[49, 3] == 0.0 and lut[50, 3] == np.float32(50 / 255.0)


def test_maps_pack_roundtrip(tmp_path):
    from monohair_amd import mapspack

    rng = np.random.default_rng(0)
    views = ["a", "b", "c"]
    o, c, m = (rng.integers(0, 256, (3, 20, 12)).astype(np.uint8) for _ in range(3))
    d = rng.random((3, 20, 12)).astype(np.float32)
    path = str(tmp_path / "maps.mhpk")
    mapspack.write_pack(path, views, {v: o[i] for i, v in enumerate(views)}, c, m, d)
    got = mapspack.read_pack(path)
    assert got["views"] == views and (got["H"], got["W"]) == (20, 12)
    for name, want in (("ori", o), ("conf", c), ("mask", m), ("depth", d)):
        assert np.array_equal(np.asarray(got[name]), want) and got[name].dtype == want.dtype
    sub = mapspack.read_pack(path, views=["c", "a"])
    assert sub["views"] == ["c", "a"] and np.array_equal(sub["ori"][0], o[2]) and np.array_equal(sub["depth"][1], d[0])
    with pytest.raises(KeyError):
        mapspack.read_pack(path, views=["zz"])
    bad = tmp_path / "bad.mhpk"
    bad.write_bytes(b"not a pack at all")
    with pytest.raises(ValueError):
        mapspack.read_pack(str(bad))
    with pytest.raises(ValueError):
        mapspack.write_pack(path, views, o, c, m[:2], d)


def test_sparse_mat_writer_equals_savemat(tmp_path):
    """The MAT-v5 files written from the occupied voxels load to the same arrays as the reference's dense
    scipy.io.savemat path (PMVO.py:753-764), duplicates resolved like its fancy assignments (last wins)."""
    import scipy.io

    from monohair_amd import pmvo_utils as U

    rng = np.random.default_rng(0)
    g = [40, 36, 28]
    v = np.stack([rng.integers(0, g[i], 500) for i in range(3)], 1)
    v[-5:] = v[:5]
    o = rng.normal(size=(500, 3)).astype(np.float32)
    a, b = tmp_path / "sparse", tmp_path / "dense"
    a.mkdir(), b.mkdir()
    U.save_ori_occ_mat_sparse(str(a), g, v, o)
    occ, ori = U.dense_from_sparse(g, v, o)
    assert occ.sum() == len(np.unique(v, axis=0)) and np.array_equal(ori[tuple(v[0])], o[-5].astype(np.float64))
    U.save_ori_occ_mat(str(b), occ, ori)
    for f, k in (("Ori3D.mat", "Ori"), ("Occ3D.mat", "Occ")):
        x, y = scipy.io.loadmat(a / f)[k], scipy.io.loadmat(b / f)[k]
        assert x.dtype == np.float64 and x.shape == y.shape and np.array_equal(x, y)
        assert os.path.getsize(a / f) == os.path.getsize(b / f)
    assert U.get_ground_truth_3D_ori(str(a / "Ori3D.mat")).shape == (28, 36, 40, 3)
    U.save_ori_occ_mat_sparse(str(a), g, np.zeros((0, 3), int), np.zeros((0, 3)))          # empty volume
    assert scipy.io.loadmat(a / "Occ3D.mat")["Occ"].sum() == 0


def test_two_phase_mat_writer_gives_the_same_bytes(tmp_path):
    """SparseMatWriter (files created and their pages made resident early by a background thread, occupied elements stored at
    the end -- what refine() uses) writes byte for byte what save_ori_occ_mat_sparse writes, whether the pre-fault hint
    covers the occupied voxels, misses them (points elsewhere / out of the grid) or is absent."""
    from monohair_amd import pmvo_utils as U

    rng = np.random.default_rng(1)
    g = [40, 36, 28]
    vmin, vs = np.array([-0.05, -0.045, -0.035]), 0.0025
    pts = rng.uniform(-0.05, 0.05, size=(800, 3))
    x, y, z = U.p2v(pts.copy(), vmin, vs, g)
    v = np.stack([x, y, z], 1).astype(np.int64)
    v[-7:] = v[:7]
    o = rng.normal(size=(len(v), 3)).astype(np.float32)
    ref = tmp_path / "ref"
    ref.mkdir()
    U.save_ori_occ_mat_sparse(str(ref), g, v, o)
    far = rng.uniform(5, 6, size=(50, 3))
    for k, hint in enumerate((pts, np.concatenate([pts[:100], far]), far, None, np.zeros((0, 3)))):
        d = tmp_path / ("two%d" % k)
        d.mkdir()
        w = U.SparseMatWriter(str(d), g, hint, vmin, vs)
        w.finish(v, o)
        assert not w._error
        for f in ("Ori3D.mat", "Occ3D.mat"):       # (the first 116 bytes are the header text with the creation time)
            assert open(d / f, "rb").read()[116:] == open(ref / f, "rb").read()[116:], (k, f)
    w = U.SparseMatWriter(str(tmp_path / "two0"), g, pts, vmin, vs)       # left early: the files of the earlier run stay
    w.abort()
    assert sorted(os.listdir(tmp_path / "two0")) == ["Occ3D.mat", "Ori3D.mat"]
    for f in ("Ori3D.mat", "Occ3D.mat"):
        assert open(tmp_path / "two0" / f, "rb").read()[116:] == open(ref / f, "rb").read()[116:]


def test_camera_tensor_utilities_match_the_oracle():
    """Camera.projection / uv2pixel / pixel2uv / reprojection / camera2world (the reference's torch utilities,
    Camera_utils.py:38-116) on CPU tensors reproduce the oracle's restatement of the same formulas bit for bit."""
    import oracle
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list

    H, W = 120, 90
    cams = cameras_from_list(synth.make_cameras(24, H, W, scale=1.5, rings=2))
    recs = camera_records(cams)
    rng = np.random.default_rng(0)
    pts = rng.normal(0, 0.1, (200, 3)).astype(np.float32)
    for i in (0, 5, 17):
        cam = list(cams.values())[i]
        uv, z = cam.projection(torch.from_numpy(pts))
        assert uv.shape == (200, 2) and z.shape == (200,)
        pix = cam.uv2pixel(uv.clone(), [H, W], "cpu")                      # (row, col), unrounded
        _, zp, _, pixf = oracle.project_points(recs[i], pts, H, W)
        assert np.array_equal(pix.numpy(), pixf) and np.array_equal((-z / 2).numpy(), zp)
        back = cam.pixel2uv(pix.clone(), [H, W], "cpu")
        assert np.allclose(back.numpy(), uv.numpy(), atol=1e-5)
        world = cam.reprojection(uv, z, to_world=True)
        assert np.allclose(world.numpy(), pts, atol=1e-5)
        camv = cam.reprojection(uv, z, to_world=False)
        assert camv.shape == (200, 4) and np.allclose(cam.camera2world(camv[:, :3]).numpy()[:, :3], pts, atol=1e-5)
    assert list(cams.values())[0].get_projection_matrix(2.0, 3.0, 0.1, -0.2).shape == (4, 4)


def test_options_equal_the_reference_on_yaml_trees_and_command_lines(tmp_path):
    """monohair_amd.options against the reference's own options.py (tests/golden/options.json, written by
    tools/gen_golden_options.py from the imported reference): `_parent_` chains and lists of parents, nested overrides,
    the --key=value / --flag / --flag! / --key= grammar with yaml-typed values, the seed -> run-name rule."""
    import json
    import os

    from conftest import GOLDEN
    from monohair_amd import options

    fx = json.load(open(os.path.join(GOLDEN, "options.json")))
    for k, run in enumerate(fx["runs"]):
        d = tmp_path / ("t%d" % k)
        d.mkdir()
        for fn, text in fx["trees"][run["tree"]].items():
            (d / fn).write_text(text.replace("{DIR}", str(d)))
        opt = options.set(options.parse_arguments(["--yaml=%s" % (d / "case")] + run["argv"]))
        got = options.to_dict(opt)
        want = dict(run["expect"])
        for key in ("yaml", "device"):       # the path of this run; "cuda:<gpu>" where a GPU is visible
            got.pop(key, None)
            want.pop(key, None)
        assert got == want, (run["tree"], run["argv"])


def test_small_file_boundary_helpers_equal_the_reference(tmp_path):
    """voxel <-> world, the .mat readers and the .hair reader / writers against the reference's own functions
    (tests/golden/utils_small.npz, tools/gen_golden_utils.py): arrays and file bytes."""
    import os

    import scipy.io
    import torch

    from conftest import GOLDEN
    from monohair_amd import pmvo_utils as U

    z = np.load(os.path.join(GOLDEN, "utils_small.npz"))
    assert np.array_equal(U.voxel_to_points(torch.from_numpy(z["vox_in"].copy())).numpy(), z["vox_to_points"])
    assert np.array_equal(U.points_to_voxel(torch.from_numpy(z["pts_in"].copy())).numpy(), z["points_to_voxel"])
    scipy.io.savemat(str(tmp_path / "Occ3D.mat"), {"Occ": z["mat_occ"]})
    scipy.io.savemat(str(tmp_path / "Ori3D.mat"), {"Ori": z["mat_ori"]})
    for flip in (0, 1):
        occ = U.get_ground_truth_3D_occ(str(tmp_path / "Occ3D.mat"), flip=bool(flip))
        ori = U.get_ground_truth_3D_ori(str(tmp_path / "Ori3D.mat"), flip=bool(flip))
        assert occ.dtype == z["occ_flip%d" % flip].dtype and np.array_equal(occ, z["occ_flip%d" % flip])
        assert ori.dtype == z["ori_flip%d" % flip].dtype and np.array_equal(ori, z["ori_flip%d" % flip])
    strands = [z["strand%d" % k] for k in range(4)]
    for t in (1, 0):
        p = str(tmp_path / ("s%d.hair" % t))
        U.save_hair_strands(p, [s.copy() for s in strands], z["bust_to_origin"], translate=bool(t))
        assert np.array_equal(np.frombuffer(open(p, "rb").read(), np.uint8), z["hair_bytes_t%d" % t])
        seg, pts = U.load_strand(p)
        assert np.array_equal(np.array(seg), z["load_seg_t%d" % t])
        assert pts.dtype == z["load_pts_t%d" % t].dtype and np.array_equal(pts, z["load_pts_t%d" % t])
    p = str(tmp_path / "w.hair")
    U.write_strand(np.concatenate(strands, 0), p, [len(s) for s in strands])
    assert np.array_equal(np.frombuffer(open(p, "rb").read(), np.uint8), z["write_strand_bytes"])


def test_camera_tensor_utilities_equal_the_reference():
    """Camera.projection / uv2pixel / pixel2uv / reprojection / camera2world against the reference's own Camera class
    (tests/golden/utils_small.npz).  Element-wise steps are bit-identical; the matrix products go through the host's
    BLAS, whose last bit depends on the CPU the fixture was written on: 1e-6."""
    import os

    from conftest import GOLDEN
    from monohair_amd.camera import cameras_from_list

    z = np.load(os.path.join(GOLDEN, "utils_small.npz"))
    H, W = 120, 90
    cams = cameras_from_list([dict(file="v%d" % i, pose=z["cam_pose_c2w"][i].tolist(), ndc_prj=z["cam_ndc"][i].tolist())
                              for i in range(len(z["cam_pose_c2w"]))])
    P = torch.from_numpy(z["cam_points"])
    for i in z["cam_views"]:
        i = int(i)
        cam = list(cams.values())[i]
        uv, zz = cam.projection(P)
        assert np.allclose(uv.numpy(), z["cam%d_uv" % i], rtol=0, atol=1e-6)
        assert np.allclose(zz.numpy(), z["cam%d_z" % i], rtol=0, atol=1e-6)
        ref_uv = torch.from_numpy(z["cam%d_uv" % i])
        pix = cam.uv2pixel(ref_uv.clone(), [H, W], "cpu")
        assert np.array_equal(pix.numpy(), z["cam%d_pix" % i])                     # element-wise: exact
        back = cam.pixel2uv(torch.from_numpy(z["cam%d_pix" % i]).clone(), [H, W], "cpu")
        assert np.array_equal(back.numpy(), z["cam%d_uvback" % i])
        ref_z = torch.from_numpy(z["cam%d_z" % i])
        assert np.allclose(cam.reprojection(ref_uv, ref_z, to_world=True).numpy(), z["cam%d_world" % i], rtol=0, atol=1e-6)
        camv = cam.reprojection(ref_uv, ref_z, to_world=False)
        assert np.allclose(camv.numpy(), z["cam%d_camv" % i], rtol=0, atol=1e-6)
        c2w = cam.camera2world(torch.from_numpy(z["cam%d_camv" % i])[:, :3])
        assert np.allclose(c2w.numpy(), z["cam%d_c2w" % i], rtol=0, atol=1e-6)


def test_unseeded_options_before_process_group_init_agree_across_ranks(tmp_path, monkeypatch):
    """seed: null under a multi-rank launcher, options.set() called BEFORE init_process_group (tools, infer_inner): every
    rank derives the same run-name suffix and numpy seed from the launcher's rendezvous environment instead of failing."""
    from monohair_amd import options

    (tmp_path / "c.yaml").write_text("seed:\ncpu: true\ngpu: 0\nname: run\n")
    names, draws = [], []
    for rank in ("0", "1"):
        monkeypatch.setenv("WORLD_SIZE", "2")
        monkeypatch.setenv("RANK", rank)
        monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
        monkeypatch.setenv("MASTER_PORT", "29999")
        monkeypatch.setenv("TORCHELASTIC_RUN_ID", "abc")
        opt = options.set(options.parse_arguments(["--yaml=%s" % (tmp_path / "c")]))
        names.append(opt.name)
        draws.append(np.random.rand())
    assert names[0] == names[1] and names[0].startswith("run_") and len(names[0]) == 8 and draws[0] == draws[1]
    monkeypatch.setenv("MASTER_PORT", "30000")               # another launch: another suffix
    assert options.set(options.parse_arguments(["--yaml=%s" % (tmp_path / "c")])).name != names[0]


def test_npy_row_writer_writes_np_save_bytes(tmp_path):
    """optimize() streams optimize/*.npy out a group of chunks at a time (monohair_amd.pmvo_utils.NpyRowWriter): the file must
    be byte for byte what np.save writes for the whole array (the reference's np.save, PMVO.py:575-579)."""
    from monohair_amd.pmvo_utils import NpyRowWriter

    rng = np.random.default_rng(0)
    for arr in (rng.normal(size=(12001, 3)).astype(np.float32), rng.normal(size=(12001,)).astype(np.float32),
                rng.random(12001) > 0.5, np.zeros((0, 3), np.float32), np.zeros((1,), np.bool_)):
        np.save(tmp_path / "a.npy", arr)
        w = NpyRowWriter(str(tmp_path / "b.npy"), arr.shape, arr.dtype)
        for lo in range(0, len(arr), 5000):
            w.write(arr[lo:lo + 5000])
        w.close()
        assert (tmp_path / "a.npy").read_bytes() == (tmp_path / "b.npy").read_bytes(), (arr.shape, arr.dtype)
        assert np.array_equal(np.load(tmp_path / "b.npy"), arr)


def test_reference_host_option_is_applied_from_file_and_inline(tmp_path):
    """PMVO.py --PMVO.reference_host=<file | JSON>: the rounding facts of the host the reference runs on become context options
    (INTEGRATION.md 3.1); the probe prints exactly that JSON."""
    import json
    import subprocess
    import sys

    sys.path.insert(0, ROOT)
    import importlib

    drv = importlib.import_module("PMVO")

    class Rec:
        def __init__(self):
            self.calls = []

        def set_option(self, k, v):
            self.calls.append((k, v))

    r = Rec()
    drv.apply_reference_host(r, None)
    assert r.calls == []
    drv.apply_reference_host(r, '{"reproject_fma_min_cols": 14223, "sum_block": 32, "threads": 4}')
    assert r.calls == [("reproject_fma_min_cols", 14223), ("sum_block", 32)]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_mkl_forms.py"), "--emit-options", "--threads", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    host = json.loads(out.stdout.strip().splitlines()[-1])
    assert host["threads"] == 1 and host["reproject_fma_min_cols"] == 2 ** 31 - 1 and host["sum_block"] == 32
    f = tmp_path / "host.json"
    f.write_text(json.dumps(host))
    r2 = Rec()
    drv.apply_reference_host(r2, str(f))
    assert ("reproject_fma_min_cols", 2 ** 31 - 1) in r2.calls
