"""CPU-only: the host-built Gabor kernels equal the reference's gabor_fn output (bit-exact where the goldens
were generated; 2e-7 elsewhere: torch's vectorised CPU sin/cos/exp are host-CPU dependent in the last bit)."""
import os

import numpy as np

from conftest import GOLDEN


def test_gabor_bank_matches_reference_kernels():
    from monohair_amd.gabor import gabor_bank, orientation_table

    z = np.load(os.path.join(GOLDEN, "gabor.npz"))
    b = gabor_bank()
    assert b.shape == (180, 17, 17) and b.dtype == np.float32
    assert np.allclose(b, z["bank"], rtol=0, atol=2e-7)
    # theta table == the radians the reference returns for each index
    th = orientation_table().numpy()
    ref = np.unique(z["noise_best"])
    assert np.all(np.isin(ref, th))


def test_dog_prefilter_definition():
    from monohair_amd.gabor import difference_of_gaussians

    img = (np.arange(64 * 48).reshape(64, 48) % 255).astype(np.uint8)
    d = difference_of_gaussians(img, 0.4, 10)
    assert d.shape == img.shape and d.dtype == np.float64 and abs(d.mean()) < 0.05
    # constant image -> exactly zero band-pass response
    assert np.abs(difference_of_gaussians(np.full((32, 32), 77, np.uint8), 0.4, 10)).max() < 1e-12
