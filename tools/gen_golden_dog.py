"""Golden vectors for the DoG prefilter of calculate_orientation (GaborFilter.py:190-192) from the REAL scikit-image:
    /opt/conda/bin/python3.9 tools/gen_golden_dog.py
The build image carries a conda tree with scikit-image 0.18.3 / scipy 1.7.1 for its python3.9 (the system python has no
scikit-image and there is no network for 0.23.2, the reference's pin).  difference_of_gaussians is the same three calls in
both versions -- img_as_float, two scipy.ndimage gaussian filters (mode 'nearest', truncate 4.0), a subtraction -- so
these vectors pin what a restatement can get wrong: the uint8 -> float conversion (a multiplication by 1/255 in float64,
skimage/util/dtype.py, not a division), the sigma / truncate handling and the output dtype.
Writes tests/golden/dog.npz (inputs are regenerated from the seeds by the test)."""
import os
import sys

import numpy as np
import scipy
import skimage
from skimage.filters import difference_of_gaussians
from skimage.util import img_as_float

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def images():
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:96, 0:80]
    stripes = (127.5 + 120 * np.sin(0.9 * xx + 0.35 * yy)).astype(np.uint8)         # hair-like texture
    noise = rng.randint(0, 256, (73, 61)).astype(np.uint8)                          # every code, odd sizes
    ramp = (np.arange(40 * 300).reshape(40, 300) % 256).astype(np.uint8)            # wider than the 10-sigma kernel
    small = rng.randint(0, 256, (9, 7)).astype(np.uint8)                            # smaller than either kernel
    codes = np.arange(256, dtype=np.uint8).reshape(16, 16)                          # the conversion of all 256 codes
    f32 = rng.rand(50, 45).astype(np.float32)                                       # a float image is not rescaled
    return dict(stripes=stripes, noise=noise, ramp=ramp, small=small, codes=codes, f32=f32)


def main():
    out = {"versions": np.array("scikit-image %s, scipy %s, numpy %s, python %s" % (
        skimage.__version__, scipy.__version__, np.__version__, sys.version.split()[0]))}
    for name, img in images().items():
        d = difference_of_gaussians(img, 0.4, 10)
        out["in_" + name] = img
        out["dog_" + name] = d
        print(name, img.shape, img.dtype, "->", d.dtype, float(d.sum()))
    out["as_float_codes"] = img_as_float(np.arange(256, dtype=np.uint8))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dog.npz"), **out)


if __name__ == "__main__":
    main()
