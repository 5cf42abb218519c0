"""The Gabor STAGE on the device: mh_dog (csrc/dog.hip) and mh_gabor_view.
  * mh_dog == the host evaluation with scipy.ndimage bit for bit (float64), == the first device form (tensor ops), and within
    1e-15 of the real scikit-image (tests/golden/dog.npz), on sizes that are not multiples of the tiles, smaller than the
    kernel radius, one row / one column, uint8 and float64 inputs;
  * mh_gabor_view (DoG -> bank -> confidence -> 8-bit file codes, four launches) == the step-by-step path."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(96, 72), (33, 129), (5, 7), (1, 50), (50, 1), (200, 64), (131, 257), (32, 64)])
@pytest.mark.parametrize("kind", ["u8", "f64"])
def test_dog_kernel_equals_scipy_and_the_tensor_form(shape, kind):
    from monohair_amd.gabor import (difference_of_gaussians, difference_of_gaussians_device,
                                    difference_of_gaussians_torch)

    rng = np.random.default_rng(hash((shape, kind)) % 2 ** 32)
    img = rng.integers(0, 256, size=shape).astype(np.uint8) if kind == "u8" else rng.normal(size=shape)
    want = difference_of_gaussians(img, 0.4, 10)
    got = difference_of_gaussians_device(img, 0.4, 10, DEV)
    assert got.dtype == torch.float64
    assert np.array_equal(got.cpu().numpy(), want)
    assert torch.equal(got, difference_of_gaussians_torch(img, 0.4, 10, DEV))
    g32 = difference_of_gaussians_device(img, 0.4, 10, DEV, out32=True)
    assert g32.dtype == torch.float32 and torch.equal(g32, got.to(torch.float32))
    # other sigmas (radius 4 and 12; the low one may be the wider)
    for lo, hi in ((1.0, 3.0), (3.0, 1.0)):
        assert np.array_equal(difference_of_gaussians_device(img, lo, hi, DEV).cpu().numpy(),
                              difference_of_gaussians(img, lo, hi))


def test_dog_kernel_against_the_real_scikit_image():
    from monohair_amd.gabor import difference_of_gaussians_device

    zd = np.load(os.path.join(GOLDEN, "dog.npz"))
    for k in ("stripes", "noise", "ramp", "small", "codes"):
        d = difference_of_gaussians_device(zd["in_" + k], 0.4, 10, DEV).cpu().numpy()
        assert np.abs(d - zd["dog_" + k]).max() <= 1e-15, k


def test_dog_radius_limit_is_a_descriptive_error():
    from monohair_amd import _lib
    from monohair_amd.gabor import difference_of_gaussians_device

    with pytest.raises(_lib.MhError, match="radius"):
        difference_of_gaussians_device(np.zeros((8, 8), np.uint8), 0.4, 20.0, DEV)       # radius 80 > 48


@pytest.mark.parametrize("shape", [(96, 72), (41, 67), (270, 480)])
def test_gabor_view_equals_the_step_by_step_stage(shape):
    from monohair_amd.gabor import calOrientationGabor, difference_of_gaussians_device, pmvo_maps_from_gabor

    H, W = shape
    rng = np.random.default_rng(H * W)
    r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    im = (127 + 70 * np.cos(2 * np.pi * (r * 0.8 + c * 0.6) / 4.0) + rng.normal(0, 6, (H, W))).clip(0, 255).astype(np.uint8)
    gab = calOrientationGabor(device=DEV)
    idx, conf, var, k8, c8 = gab.view(im)
    dog = difference_of_gaussians_device(im, 0.4, 10, DEV, out32=True)
    idx2, conf2, var2 = gab.filter_index(dog)
    assert torch.equal(idx, idx2) and torch.equal(conf, conf2) and torch.equal(var, var2)
    _, _, k8b, c8b = pmvo_maps_from_gabor(idx2, conf2)
    assert torch.equal(k8, k8b) and torch.equal(c8, c8b)
    assert int(c8.max()) == 255                                  # the image maximum normalises to confidence 1
    # a second view through the same scratch, and a device tensor as input
    idx3, _, _, k83, _ = gab.view(torch.from_numpy(im[::-1].copy()).to(DEV))
    assert not torch.equal(idx3, idx)
    idx4, conf4, _, k84, c84 = gab.view(im)
    assert torch.equal(idx4, idx) and torch.equal(k84, k8) and torch.equal(c84, c8)


def test_gabor_view_at_1080p_equals_the_oracle():
    """The whole per-view stage at the headline size (GaborFilter.py:29-113,186-210), default
    kernels: uint8 image -> DoG (float64) -> 180-filter bank (mfma2) -> variance / confidence -> 8-bit file codes, against
    the host statement of the same chain: scipy's DoG, the C oracle's bank on ALL 1920x1080 pixels (so the image-wide maximum
    that normalises the confidence is the oracle's too), and the file-code arithmetic in numpy float32."""
    import oracle
    from monohair_amd.gabor import calOrientationGabor, difference_of_gaussians, gabor_bank

    H, W = 1920, 1080
    rng = np.random.default_rng(11)
    r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    wave = np.cos(2 * np.pi * (r * np.cos(0.002 * c) + c * np.sin(0.002 * c)) / 4.0)      # orientation drifts over the image
    im = (127 + 70 * wave * (r > 200) + rng.normal(0, 6, (H, W))).clip(0, 255).astype(np.uint8)
    gab = calOrientationGabor(device=DEV)
    assert gab.variant == "mfma2"
    idx, conf, var, k8, c8 = gab.view(im)
    dog = difference_of_gaussians(im, 0.4, 10).astype(np.float32)
    o_idx, o_conf, o_var = oracle.gabor_bank(gabor_bank(), dog)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    assert np.array_equal(var.cpu().numpy(), o_var)
    assert np.array_equal(conf.cpu().numpy(), o_conf)
    assert len(np.unique(o_idx)) > 90
    want_k8 = np.clip(o_idx, 0, 255).astype(np.uint8)
    want_c8 = np.clip(o_conf * np.float32(255.0) + np.float32(0.5), 0, 255).astype(np.uint8)
    assert np.array_equal(k8.cpu().numpy(), want_k8) and np.array_equal(c8.cpu().numpy(), want_c8)


@pytest.mark.parametrize("ranks", [2, 3])
def test_view_sharded_gabor_stage_equals_the_single_rank_stage(tmp_path, ranks):
    """SURVEY.md §8e / the north star's "views shard across the GPUs" (GaborFilter.py:231-237 is a loop over the views): 7 views
    dealt to 2 and 3 ranks (not a multiple of either; gloo ranks sharing the test GPU), one all_gather of the 2 B/px code
    planes -> every rank holds the single-rank codes and maps byte for byte, and batch_generate's files are the single-rank
    files byte for byte.  The single-rank codes themselves are checked against the oracle per view."""
    import subprocess
    import sys

    import oracle
    from conftest import ROOT
    from monohair_amd.gabor import difference_of_gaussians, gabor_bank

    helper = os.path.join(ROOT, "tests", "gabor_ranks_helper.py")
    env = dict(os.environ, PYTHONPATH=ROOT)
    one, many = tmp_path / "one", tmp_path / "many"
    r = subprocess.run([sys.executable, helper, "--out", str(one), "--views", "7"], cwd=ROOT, env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    env2 = dict(env, MH_DIST_BACKEND="gloo", MH_DEVICE_OVERRIDE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(29630 + ranks), helper, "--out", str(many),
                        "--views", "7"], cwd=ROOT, env=env2, stdin=subprocess.DEVNULL, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    want = np.load(one / "codes_rank0.npz")
    want_maps = np.load(one / "maps_rank0.npz")
    assert want["k8"].shape == (7, 150, 100)
    for rk in range(ranks):
        got = np.load(many / ("codes_rank%d.npz" % rk))
        assert np.array_equal(got["k8"], want["k8"]) and np.array_equal(got["c8"], want["c8"]), rk
        gm = np.load(many / ("maps_rank%d.npz" % rk))
        assert np.array_equal(gm["ori"], want_maps["ori"]) and np.array_equal(gm["conf"], want_maps["conf"]), rk
    for sub in ("best_ori", "conf", "Ori"):
        names = sorted(os.listdir(one / "files" / sub))
        assert names == sorted(os.listdir(many / "files" / sub)) and len(names) == 7
        for n in names:
            assert (one / "files" / sub / n).read_bytes() == (many / "files" / sub / n).read_bytes(), (sub, n)
    # and the single-rank codes are the oracle's (two of the views)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gabor_ranks_helper import images

    ims = images(7)
    for v in (0, 4):
        dog = difference_of_gaussians(ims[v], 0.4, 10).astype(np.float32)
        o_idx, o_conf, _ = oracle.gabor_bank(gabor_bank(), dog)
        assert np.array_equal(want["k8"][v], np.clip(o_idx, 0, 255).astype(np.uint8))
        assert np.array_equal(want["c8"][v], np.clip(o_conf * np.float32(255) + np.float32(0.5), 0, 255).astype(np.uint8))
