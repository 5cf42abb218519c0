"""GPU: the drop-in entry point `PMVO.py` end to end through the real file loaders on a synthetic capture written
in the reference's on-disk layout, both passes (exterior, then --PMVO.optimize= resume)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_pmvo_cli_end_to_end(tmp_path):
    import scipy.io

    from monohair_amd import synth

    data = tmp_path / "data"
    synth.write_case(str(data), "synthetic_sphere", V=24, H=240, W=136, res=32)
    common = [sys.executable, os.path.join(ROOT, "PMVO.py"), "--yaml=configs/reconstruct/synthetic_sphere",
              "--data.root=%s" % data, "--data.image_size=[240,136]", "--PMVO.patch_size=3", "--name=t1"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run(common, cwd=ROOT, env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = data / "synthetic_sphere" / "output" / "t1"
    for f in ("options.yaml", "optimize/surface.npy", "optimize/filter_unvisible.npy", "optimize/select_p.npy",
              "optimize/select_o.npy", "optimize/min_loss.npy", "optimize/high_conf_index.npy", "refine/select_p.npy",
              "refine/select_o.npy", "refine/min_loss.npy", "refine/filter_unvisible.npy",
              "refine/filter_unvisible_ori.npy", "refine/Ori3D.mat", "refine/Occ3D.mat"):
        assert (out / f).exists(), f
    occ = scipy.io.loadmat(out / "refine" / "Occ3D.mat")["Occ"]
    ori = scipy.io.loadmat(out / "refine" / "Ori3D.mat")["Ori"]
    assert occ.shape == (256, 256, 192) and ori.shape == (256, 256, 576) and occ.sum() > 500
    # known answer: fitted directions follow the meridian tangent field of the sphere
    nz = np.argwhere(occ != 0)
    o = np.stack([ori[nz[:, 0], nz[:, 1], c * 192 + nz[:, 2]] for c in range(3)], 1)
    p = np.stack([nz[:, 1], nz[:, 0], nz[:, 2]], 1) * 0.0025 + np.array([-0.32, -0.32, -0.24])   # [Y,X,Z] -> xyz
    p[:, 1:] *= -1
    n = p / np.linalg.norm(p, axis=1, keepdims=True)
    t = -np.array([0, 1.0, 0])[None] + n[:, 1:2] * n
    ok = np.linalg.norm(t, axis=1) > 0.3
    t = t[ok] / np.linalg.norm(t[ok], axis=1, keepdims=True)
    cosv = np.abs((t * o[ok]).sum(1)) / np.maximum(np.linalg.norm(o[ok], axis=1), 1e-9)
    assert np.median(cosv) > 0.97, np.median(cosv)
    # second pass: resume from optimize/*.npy (PMVO.py:874-880), output under full/ when infer_inner is set
    np.save(data / "synthetic_sphere" / "ours" / "raw.npy",
            np.concatenate([p[:50], o[:50], np.ones((50, 1))], 1).astype(np.float32))
    r = subprocess.run(common + ["--PMVO.optimize=", "--PMVO.infer_inner"], cwd=ROOT, env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (out / "full" / "Ori3D.mat").exists() and (out / "full" / "coarse.npy").exists()


@pytest.mark.parametrize("ranks,refine_shard,res", [(2, "0", 32), (2, "1", 32), (3, "1", 64), (8, "1", 64)])
def test_two_ranks_give_the_single_rank_volume_bit_for_bit(tmp_path, ranks, refine_shard, res):
    """the path shards by points and the voxel fit by disjoint slabs + ONE exchange, so an N-rank run must
    reproduce the 1-rank outputs bit for bit.  The ranks share the single test GPU (gloo; RCCL refuses two ranks on
    one device), which exercises map_chunks, the all_gather and the volume exchange with the real kernels.
    refine_shard "1": refine's smoothing loop and shell stage sharded over the ranks as well (what the nccl backend does by
    default: every rank owns a slice of every 5000-point chunk, one in-place all_gather per chunk); 3 ranks on the larger
    case: several chunks, slices that do not divide a chunk; 8 ranks: the rank count of BASELINE.json's multi-GPU
    configurations (more ranks than chunks in some stages, eight x-slabs of the volume)."""
    import scipy.io

    from monohair_amd import synth

    data = tmp_path / "data"
    synth.write_case(str(data), "synthetic_sphere", V=24, H=240, W=136, res=res)
    base = ["--yaml=configs/reconstruct/synthetic_sphere", "--data.root=%s" % data, "--data.image_size=[240,136]",
            "--PMVO.patch_size=3"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "PMVO.py")] + base + ["--name=one"], cwd=ROOT, env=env,
                        stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    env2 = dict(env, MH_DIST_BACKEND="gloo", MH_DEVICE_OVERRIDE="0", MH_REFINE_SHARD=refine_shard)
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                         "--master-addr", "127.0.0.1", "--master-port", str(29561 + ranks + int(refine_shard)),
                         os.path.join(ROOT, "PMVO.py")] + base +
                        ["--name=two"], cwd=ROOT, env=env2, stdin=subprocess.DEVNULL, capture_output=True, text=True,
                        timeout=900)
    assert r2.returncode == 0, r2.stdout[-3000:] + r2.stderr[-3000:]
    out = data / "synthetic_sphere" / "output"
    for f in ("optimize/select_p.npy", "optimize/select_o.npy", "optimize/min_loss.npy", "optimize/surface.npy",
              "refine/select_o.npy", "refine/min_loss.npy", "refine/filter_unvisible_ori.npy"):
        a, b = np.load(out / "one" / f), np.load(out / "two" / f)
        assert np.array_equal(a, b, equal_nan=True), f
    for f, k in (("refine/Ori3D.mat", "Ori"), ("refine/Occ3D.mat", "Occ")):
        a, b = scipy.io.loadmat(out / "one" / f)[k], scipy.io.loadmat(out / "two" / f)[k]
        assert np.array_equal(a, b), f


def test_u8_upload_and_maps_pack_equal_the_float_loaders(tmp_path):
    """The GPU decode of the 8-bit files (PMVO.from_u8) must leave the same resident maps as the reference's
    float64 host decode + constructor, and a run from a maps pack must write the same files as a run from the tree."""
    import torch

    from monohair_amd import pmvo_utils as U
    from monohair_amd import synth
    from monohair_amd.camera import load_cam, parsing_camera
    from monohair_amd.pmvo import PMVO

    data = tmp_path / "data"
    base = synth.write_case(str(data), "synthetic_sphere", V=24, H=240, W=136, res=32)
    camera = parsing_camera(load_cam(os.path.join(base, "ours/cam_params.json")), os.path.join(base, "capture_images"))
    p = lambda d: os.path.join(base, d)   # noqa: E731
    Ori, Conf = U.Load_Ori_And_Conf(camera, p("best_ori"), p("conf"))
    a = PMVO(camera, U.load_depth(camera, p("render_depth")), Ori, Conf, U.load_mask(camera, p("hair_mask")),
             device="cuda:0", image_size=[240, 136], patch_size=5, conf_threshold=0.15)
    o8, c8, m8 = U.load_maps_u8(camera, p("best_ori"), p("conf"), p("hair_mask"))
    b = PMVO.from_u8(camera, U.load_depth_plane(camera, p("render_depth")), o8, c8, m8, device="cuda:0",
                     image_size=[240, 136], patch_size=5, conf_threshold=0.15)
    pts = synth.candidate_points(res=32, seed=3)[:3000]
    for pm in (a, b):
        pm.Compute_Visible_and_Ori(pts)
    for name in ("visible", "Ori", "Conf", "mask", "Ori_patch", "Conf_patch"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    ra, rb = a.forward(pts[:500]), b.forward(pts[:500])
    assert all(torch.equal(x, y) or (x != x).equal(y != y) for x, y in zip(ra[1:], rb[1:]))

    common = [sys.executable, os.path.join(ROOT, "PMVO.py"), "--yaml=configs/reconstruct/synthetic_sphere",
              "--data.root=%s" % data, "--data.image_size=[240,136]", "--PMVO.patch_size=3"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    for extra in (["--name=tree"], ["--name=pack", "--data.maps_pack=maps.mhpk"],
                  ["--name=pack2", "--data.maps_pack=maps.mhpk"]):        # pack2 reads the pack written by `pack`
        r = subprocess.run(common + extra, cwd=ROOT, env=env, stdin=subprocess.DEVNULL, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert os.path.getsize(p("maps.mhpk")) > 24 * 240 * 136 * 7
    out = data / "synthetic_sphere" / "output"
    for f in ("optimize/select_p.npy", "optimize/select_o.npy", "optimize/min_loss.npy", "refine/select_o.npy",
              "refine/min_loss.npy"):
        for other in ("pack", "pack2"):
            assert np.array_equal(np.load(out / "tree" / f), np.load(out / other / f), equal_nan=True), (f, other)


def test_infer_inner_cli_renders_the_segment_images(tmp_path):
    """infer_inner.py (the reference's caller of the second pass, :30-90): exterior pass -> traced segments ->
    imgs/<view>/{bust_depth,undirectional_map,mask,hair_depth}.png + refine/render_segments.hair; with ours/raw.npy present
    the second PMVO pass runs and writes full/."""
    from PIL import Image

    from monohair_amd import synth
    from monohair_amd.pmvo_utils import load_strand

    data = tmp_path / "data"
    base = synth.write_case(str(data), "synthetic_sphere", V=24, H=240, W=136, res=32)
    env = dict(os.environ, PYTHONPATH=ROOT)
    common = ["--yaml=configs/reconstruct/synthetic_sphere", "--data.root=%s" % data, "--data.image_size=[240,136]",
              "--PMVO.patch_size=3", "--name=t1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "PMVO.py")] + common, cwd=ROOT, env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    cam = os.path.join(base, "ours", "cam_params.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "infer_inner.py")] + common +
                       ["--camera_path=%s" % cam, "--infer_inner.run_mvs="], cwd=ROOT, env=env, stdin=subprocess.DEVNULL,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = data / "synthetic_sphere" / "output" / "t1"
    segs, pts = load_strand(str(out / "refine" / "render_segments.hair"))
    assert len(segs) >= 1 and pts.shape[0] == sum(segs)     # (the 32^3 test volume is sparse: few, short segments)
    views = sorted(os.listdir(os.path.join(base, "imgs")))
    assert len(views) == 24
    frac = []
    for v in views[:6]:
        ims = {n: np.array(Image.open(os.path.join(base, "imgs", v, n + ".png"))) for n in
               ("bust_depth", "undirectional_map", "mask", "hair_depth")}
        assert all(im.shape == (1280, 720, 3) and im.dtype == np.uint8 for im in ims.values())
        m = ims["mask"][..., 0] == 255
        frac.append(m.sum())
        assert set(np.unique(ims["mask"]).tolist()) <= {0, 255}
        assert (ims["undirectional_map"][m][:, 2] == 0).all() and (ims["undirectional_map"][~m] == 0).all()
        assert (ims["hair_depth"][m][:, 0] < 255).all()                   # depth/2 of a point ~0.7 m away
    assert max(frac) > 0                                                  # some view sees the traced segments
    # second pass through infer_inner.py once DeepMVSHair's output exists
    p = np.load(out / "refine" / "select_p.npy")
    o = np.load(out / "refine" / "select_o.npy")
    np.save(os.path.join(base, "ours", "raw.npy"), np.concatenate([p[:50] * 0.5, o[:50], np.ones((50, 1))], 1).astype(np.float32))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "infer_inner.py")] + common +
                       ["--camera_path=%s" % cam, "--infer_inner.render_data="], cwd=ROOT, env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert (out / "full" / "Ori3D.mat").exists() and (out / "full" / "coarse.npy").exists()
