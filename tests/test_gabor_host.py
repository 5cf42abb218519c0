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


def test_dog_prefilter_equals_scikit_image():
    """difference_of_gaussians against the real scikit-image (tests/golden/dog.npz, tools/gen_golden_dog.py): the
    uint8 -> float conversion bit for bit (a multiplication by 1/255, not a division), a float32 image bit for bit
    (it stays float32), the filtered uint8 images to 1e-15: the Gaussian weights go through numpy's exp, whose last bit
    differs between the numpy that wrote the fixture (1.26) and the one that runs here (values are O(1), float64)."""
    import os

    from conftest import GOLDEN
    from monohair_amd.gabor import _img_as_float, difference_of_gaussians

    z = np.load(os.path.join(GOLDEN, "dog.npz"))
    assert np.array_equal(_img_as_float(np.arange(256, dtype=np.uint8)), z["as_float_codes"])
    seen = 0
    for k in z.files:
        if not k.startswith("in_"):
            continue
        got, want = difference_of_gaussians(z[k], 0.4, 10), z["dog_" + k[3:]]
        assert got.dtype == want.dtype and got.shape == want.shape
        if z[k].dtype == np.float32:
            assert np.array_equal(got, want)
        else:
            assert np.abs(got - want).max() <= 1e-15
            # the float32 image the Gabor bank receives (ToTensor + .type(torch.float), GaborFilter.py:197)
            assert np.mean(got.astype(np.float32) == want.astype(np.float32)) > 0.999
        seen += 1
    assert seen == 6
