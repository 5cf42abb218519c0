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
