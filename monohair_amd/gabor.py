"""Per-view Gabor orientation / confidence maps -- the host-side mirror of the reference's
`preprocess_capture_data/GaborFilter.py` (calOrientationGabor :16-145, calculate_orientation :164-224,
batch_generate :231-237) on top of the HIP kernel mh_gabor_bank (monohair_amd/csrc/gabor.hip).

The 180 kernels are built on the host with the same CPU torch ops as gabor_fn (so the bank is the reference's
bits) and installed once with mh_gabor_set_bank; the 1.49 GB response stack of the reference never exists.
File IO uses PIL; the difference-of-Gaussians prefilter is scikit-image's documented definition evaluated with
scipy.ndimage (third-party arithmetic, unpinned -- SURVEY.md §8c)."""
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from .pmvo_utils import _ctx_for

NUM_KERNELS = 180
KSIZE = 17


def gabor_fn(kernel_size, theta, sigma_x=1.8, sigma_y=2.4, Lambda=4.0, phase=0.0):
    """One real Gabor kernel [k,k] (GaborFilter.py:115-145): taps x,y in {-8.5..7.5} (x <-> row offset),
    x_t = x cos t + y sin t, y_t = -x sin t + y cos t, exp(-.5 (x_t^2/sx^2 + y_t^2/sy^2)) cos(2 pi x_t / L + psi);
    float32 CPU torch ops in the reference's order."""
    half = kernel_size // 2
    t = torch.ones(1) * theta
    sx, sy = torch.ones(1) * sigma_x, torch.ones(1) * sigma_y
    lam, psi = torch.ones(1) * Lambda, torch.ones(1) * phase
    r = torch.arange(-half, half + 1).float() - 0.5
    x = r.view(-1, 1).repeat(1, kernel_size)
    y = r.view(1, -1).repeat(kernel_size, 1)
    x_t = x * torch.cos(t) + y * torch.sin(t)
    y_t = -x * torch.sin(t) + y * torch.cos(t)
    return torch.exp(-.5 * (x_t ** 2 / sx ** 2 + y_t ** 2 / sy ** 2)) * torch.cos(2 * math.pi * x_t / lam + psi)


_BANK = None


def gabor_bank():
    """[180,17,17] float32: theta_k = pi*k/180 (GaborFilter.py:32-34)."""
    global _BANK
    if _BANK is None:
        _BANK = torch.stack([gabor_fn(KSIZE, math.pi * k / NUM_KERNELS) for k in range(NUM_KERNELS)]).numpy()
    return _BANK


def orientation_table():
    """theta_k as fp32 exactly as the reference forms best_orientTensor (index * pi / 180, GaborFilter.py:51)."""
    return (torch.arange(NUM_KERNELS).float() * math.pi / NUM_KERNELS)


class calOrientationGabor:
    def __init__(self, channel_in=1, channel_out=1, stride=1, device=None, bank=None, variant="mfma2"):
        """bank: optional [180,17,17] kernels to install instead of gabor_bank() (torch's CPU sin/cos/exp differ
        in the last bit between host CPU types; tests pin the reference's own kernels this way)."""
        self.numKernels = NUM_KERNELS
        self.clamp_confidence_low = 0.0
        self.clamp_confidence_high = 0.2
        if not torch.cuda.is_available():
            raise _lib.MhError("calOrientationGabor needs a ROCm GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._ctx = _ctx_for(self.device)
        bank = gabor_bank() if bank is None else np.asarray(bank)
        bank = np.ascontiguousarray(bank.reshape(NUM_KERNELS, KSIZE * KSIZE), dtype=np.float32)
        _lib.check(_lib.lib().mh_gabor_set_bank(self._ctx, bank.ctypes.data_as(ctypes.c_void_p)), "mh_gabor_set_bank")
        self.set_variant(variant)
        th = orientation_table()
        self._theta = th.to(self.device)
        self._sin = torch.sin(th).to(self.device)      # CPU transcendental values, gathered on the device
        self._cos = torch.cos(th).to(self.device)

    def set_variant(self, variant):
        """'mfma2' (default): the bank as an im2col contraction on v_mfma_f32_32x32x2_f32; 'valu': the direct form on
        v_pk_fma_f32, kept as the cross-check.  Same bits."""
        self.variant = variant
        _lib.check(_lib.lib().mh_ctx_set_option(self._ctx, b"gabor_variant",
                                                {"valu": 0, "mfma2": 3}[variant]),
                   "mh_ctx_set_option")

    def cuda(self):
        return self

    def filter_index(self, image):
        """image [H,W] float32 device tensor -> (orient index int32 [H,W], conf [H,W], variance [H,W])."""
        image = image.to(self.device).type(torch.float).contiguous()
        H, W = image.shape
        idx = torch.empty((H, W), dtype=torch.int32, device=self.device)
        conf = torch.empty((H, W), dtype=torch.float32, device=self.device)
        var = torch.empty((H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mh_gabor_bank(self._ctx, _lib.ptr(image), H, W, _lib.ptr(idx), _lib.ptr(conf),
                                                _lib.ptr(var), _lib.stream_ptr()), "mh_gabor_bank")
        return idx, conf, var

    def view(self, gray_u8, codes=True, low_sigma=0.4, high_sigma=10):
        """One view of the stage, device to device (mh_gabor_view): gray uint8 [H,W] (numpy or device tensor) -> DoG ->
        bank -> (orient index int32, conf fp32, variance fp32, best_ori code uint8, conf code uint8); the codes are what the
        reference writes to best_ori/<view> and conf/<view> (None with codes=False).  Four launches, no tensor op."""
        g = gray_u8 if torch.is_tensor(gray_u8) else torch.from_numpy(np.ascontiguousarray(gray_u8))
        g = g.to(self.device).contiguous()
        assert g.dtype == torch.uint8 and g.dim() == 2
        self._install_bank(1.8, 2.4, 4.0)
        H, W = g.shape
        L = _lib.lib()
        key = (H, W, torch.cuda.current_stream(self.device).cuda_stream)
        pool = self.__dict__.setdefault("_view_scratches", {})       # one scratch per (size, stream), reused across views
        if key not in pool:
            if len(pool) >= 8:
                pool.clear()
            pool[key] = torch.empty((L.mh_gabor_view_scratch_bytes(H, W),), dtype=torch.uint8, device=self.device)
        scratch = pool[key]
        idx = torch.empty((H, W), dtype=torch.int32, device=self.device)
        conf = torch.empty((H, W), dtype=torch.float32, device=self.device)
        var = torch.empty((H, W), dtype=torch.float32, device=self.device)
        k8 = torch.empty((H, W), dtype=torch.uint8, device=self.device) if codes else None
        c8 = torch.empty((H, W), dtype=torch.uint8, device=self.device) if codes else None
        w0, r0, w1, r1 = _dog_weights(low_sigma, high_sigma)
        with torch.cuda.device(self.device):
            _lib.check(L.mh_gabor_view(self._ctx, _lib.ptr(g), H, W, w0.ctypes.data_as(ctypes.c_void_p), r0,
                                       w1.ctypes.data_as(ctypes.c_void_p), r1, _lib.ptr(scratch), _lib.ptr(idx),
                                       _lib.ptr(conf), _lib.ptr(var), _lib.ptr(k8), _lib.ptr(c8), _lib.stream_ptr()),
                       "mh_gabor_view")
        return idx, conf, var, k8, c8

    def gabor_fn(self, kernel_size, channel_in, channel_out, theta, sigma_x, sigma_y, Lambda, phase=0.):
        """GaborFilter.py:115-145 with the reference's signature -> [channel_out, channel_in, k, k] on the device"""
        th = torch.as_tensor(theta, dtype=torch.float32).reshape(-1).cpu()
        ks = [gabor_fn(kernel_size, float(th[min(i, len(th) - 1)]), sigma_x, sigma_y, Lambda, phase)
              for i in range(channel_out)]
        return torch.stack(ks)[:, None].repeat(1, channel_in, 1, 1).to(self.device)

    def _install_bank(self, sigma_x, sigma_y, Lambda):
        key = (float(sigma_x), float(sigma_y), float(Lambda))
        if getattr(self, "_bank_key", (1.8, 2.4, 4.0)) != key:
            bank = torch.stack([gabor_fn(KSIZE, math.pi * k / NUM_KERNELS, *key) for k in range(NUM_KERNELS)]).numpy()
            bank = np.ascontiguousarray(bank.reshape(NUM_KERNELS, KSIZE * KSIZE), dtype=np.float32)
            _lib.check(_lib.lib().mh_gabor_set_bank(self._ctx, bank.ctypes.data_as(ctypes.c_void_p)), "mh_gabor_set_bank")
            self._bank_key = key

    def filter(self, image, label, threshold, variance_data, orient_data, max_resp_data, sigma_x=1.8, sigma_y=2.4,
               Lambda=4, kernel_size=17):
        """GaborFilter.py:29-94, one pass: the bank on image [1,1,H,W], then the reference's state update
        (where variance > variance_data ...), both normalisations and the clamp.  Returns (confidence, variance_data,
        orient_data), each [1,1,H,W].  (max_resp_data never reaches an output of the reference either.)"""
        if int(kernel_size) != KSIZE:
            raise NotImplementedError("the bank kernel is built for 17x17 filters (GaborFilter.py:105)")
        self._install_bank(sigma_x, sigma_y, Lambda)
        idx, _, var = self.filter_index(image[0, 0])
        best = self._theta[idx.long()][None, None]
        variance = var[None, None]
        upd = variance > variance_data.to(self.device)
        orient_data = torch.where(upd, best, orient_data.to(self.device))
        variance_data = torch.where(upd, variance, variance_data.to(self.device))
        variance_data = variance_data / torch.max(variance_data)                  # tensor divisor: IEEE division
        span = torch.tensor(self.clamp_confidence_high - self.clamp_confidence_low, dtype=torch.float32,
                            device=self.device)
        conf = ((variance_data - self.clamp_confidence_low) / span).clamp(0, 1)
        return conf, variance_data, orient_data

    def forward(self, image, label=None, iter=1, threshold=0.0):
        """GaborFilter.py:98-113.  image [1,1,H,W] -> (orientTwoChannel [1,2,H,W] = (sin,cos),
        best_orient [1,1,H,W] radians, confidence [1,1,H,W]).  `label` is unused, as in the reference.  iter=1 (what
        the pipeline uses, GaborFilter.py:236) is one fused launch; iter>1 re-filters the confidence map like the
        reference, through filter()."""
        if iter == 1:
            self._install_bank(1.8, 2.4, 4.0)
            idx, conf, _ = self.filter_index(image[0, 0])
            li = idx.long()
            conf = torch.where(conf < threshold, torch.zeros_like(conf), conf)
            best = self._theta[li]
            two = torch.stack([self._sin[li], self._cos[li]], 0)[None]
            return two, best[None, None], conf[None, None]
        H, W = image.shape[2:4]
        z = lambda: torch.zeros((1, 1, H, W), dtype=torch.float32, device=self.device)   # noqa: E731
        variance_data, orient_data, max_resp_data = z(), z(), z()
        image = image.to(self.device).type(torch.float)
        for _ in range(iter):
            conf, variance_data, orient_data = self.filter(image, label, threshold, variance_data, orient_data,
                                                           max_resp_data, sigma_x=1.8, sigma_y=2.4, Lambda=4,
                                                           kernel_size=17)
            image = conf
        conf = torch.where(conf < threshold, torch.zeros_like(conf), conf)
        oc = orient_data.cpu()             # sin/cos of the few distinct angles with the host's libm, as for iter=1
        return torch.cat([torch.sin(oc), torch.cos(oc)], dim=1).to(self.device), orient_data, conf

    __call__ = forward


def normalize(x):
    """GaborFilter.py:152-155"""
    return x / np.maximum(np.linalg.norm(x, axis=-1)[..., None], 1e-8)


def convert_numpy(tensor):
    """GaborFilter.py:157-161: [1,C,H,W] tensor -> [H,W,C] numpy"""
    return torch.squeeze(tensor, 0).permute(1, 2, 0).data.cpu().numpy()


def _img_as_float(image):
    """skimage.util.img_as_float for the inputs that occur here: uint8 -> float64 by a MULTIPLICATION with 1/255
    (skimage/util/dtype.py `_convert`: np.multiply(image, 1. / imax_in, dtype=float64) -- not a division: the two differ
    in the last bit for many codes); float32 / float64 arrays pass through with their precision."""
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return np.multiply(image, 1.0 / 255, dtype=np.float64)
    return image if image.dtype in (np.float32, np.float64) else image.astype(np.float64)


def difference_of_gaussians(image, low_sigma, high_sigma):
    """skimage.filters.difference_of_gaussians as the reference calls it (GaborFilter.py:192): img_as_float, two
    scipy.ndimage.gaussian_filter passes (mode='nearest', truncate=4.0), their difference.  Pinned to the real
    scikit-image by tests/golden/dog.npz (tools/gen_golden_dog.py)."""
    from scipy import ndimage as ndi

    img = _img_as_float(image)
    lo = ndi.gaussian_filter(img, low_sigma, mode="nearest", truncate=4.0)
    hi = ndi.gaussian_filter(img, high_sigma, mode="nearest", truncate=4.0)
    return lo - hi


def _gaussian_kernel1d(sigma, truncate=4.0):
    """scipy.ndimage's 1-D Gaussian weights (float64): radius = int(truncate*sigma + 0.5), normalised."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum(), radius


def _correlate1d_nearest(x, w, radius, axis):
    """scipy.ndimage.correlate1d of a symmetric kernel, mode='nearest', in its accumulation order
    (ni_filters.c: tmp = x[l]*w[c]; for j = -r..-1: tmp += (x[l+j] + x[l-j]) * w[j+r]) -- float64 torch ops."""
    n = x.shape[axis]
    idx = torch.arange(-radius, n + radius, device=x.device).clamp(0, n - 1)
    xp = x.index_select(axis, idx)                       # 'nearest' padding
    centre = xp.narrow(axis, radius, n)
    acc = centre * float(w[radius])
    for j in range(-radius, 0):
        acc = acc + (xp.narrow(axis, radius + j, n) + xp.narrow(axis, radius - j, n)) * float(w[j + radius])
    return acc


def difference_of_gaussians_torch(image, low_sigma, high_sigma, device):
    """The first device form (about 250 float64 tensor launches per image); kept as the cross-check of the HIP kernel
    (tests assert array_equal)."""
    img = np.asarray(image)
    x = torch.from_numpy(img).to(device)
    # img_as_float: uint8 codes are MULTIPLIED by the float64 constant 1/255 (see _img_as_float)
    x = x.to(torch.float64) * (1.0 / 255) if img.dtype == np.uint8 else x.to(torch.float64)
    out = []
    for sigma in (low_sigma, high_sigma):
        w, r = _gaussian_kernel1d(sigma)
        y = _correlate1d_nearest(x, w, r, 0)
        out.append(_correlate1d_nearest(y, w, r, 1))
    return out[0] - out[1]


_DOG_W = {}


def _dog_weights(low_sigma, high_sigma):
    """the two symmetric half kernels (float64, w[r] = centre) as contiguous host arrays for mh_dog / mh_gabor_view"""
    key = (float(low_sigma), float(high_sigma))
    if key not in _DOG_W:
        out = []
        for sigma in key:
            w, r = _gaussian_kernel1d(sigma)
            out += [np.ascontiguousarray(w[:r + 1], dtype=np.float64), int(r)]
        _DOG_W[key] = tuple(out)
    return _DOG_W[key]


def difference_of_gaussians_device(image, low_sigma, high_sigma, device, out32=False):
    """difference_of_gaussians on the GPU in float64 with scipy's operation order (separable passes along axis 0 then 1):
    mh_dog, two launches (csrc/dog.hip).  image: uint8 [H,W] (what the reference passes; scaled like skimage's
    img_as_float) or a float array, which is filtered in float64 (scikit-image keeps a float32 image in float32 -- use
    difference_of_gaussians for that); a numpy array or a device tensor.  Returns the float64 image, or with out32=True its
    float32 cast (what the bank takes)."""
    device = torch.device(device)
    if torch.is_tensor(image):
        x = image.to(device)
    else:
        x = torch.from_numpy(np.ascontiguousarray(image)).to(device)
    if x.dtype != torch.uint8:
        x = x.to(torch.float64)
    x = x.contiguous()
    H, W = x.shape
    w0, r0, w1, r1 = _dog_weights(low_sigma, high_sigma)
    L = _lib.lib()
    scratch = torch.empty((L.mh_dog_scratch_bytes(H, W),), dtype=torch.uint8, device=device)
    out = torch.empty((H, W), dtype=torch.float32 if out32 else torch.float64, device=device)
    with torch.cuda.device(device):
        _lib.check(L.mh_dog(_ctx_for(device), _lib.ptr(x), 0 if x.dtype == torch.uint8 else 1, H, W,
                            w0.ctypes.data_as(ctypes.c_void_p), r0, w1.ctypes.data_as(ctypes.c_void_p), r1,
                            _lib.ptr(scratch), None if out32 else _lib.ptr(out), _lib.ptr(out) if out32 else None,
                            _lib.stream_ptr()), "mh_dog")
    return out


_FILE_LUT = None


def pmvo_maps_from_gabor(index, conf):
    """GPU-resident hand-off Gabor -> PMVO that reproduces the reference's round trip through 8-bit image files
    (SURVEY.md Appendix A.18): best_ori/<view> stores the orientation in integer degrees (GaborFilter.py:209), conf/<view>
    stores floor(conf*255+0.5) (torchvision.save_image, :210); the loaders turn them into (sin t', cos t') with
    t' = (180-pix)/180*pi and conf = pix/255 in float64 (PMVO_utils.py:265-272), cast to fp32 by PMVO.__init__.
    index: int32 [H,W] (degrees), conf: fp32 [H,W] in [0,1] -> (ori [H,W,2] fp32, conf [H,W] fp32) on the device.
    Also returns the two uint8 planes (2 B/px: what travels between GPUs)."""
    global _FILE_LUT
    if _FILE_LUT is None:
        pix = np.arange(256, dtype=np.float64)
        th = (180 - pix) / 180 * math.pi
        _FILE_LUT = (torch.from_numpy(np.stack([np.sin(th), np.cos(th)], -1).astype(np.float32)),
                     torch.from_numpy((pix / 255.0).astype(np.float32)))
    dev = index.device
    k8 = index.clamp(0, 255).to(torch.uint8)
    c8 = (conf * 255 + 0.5).clamp(0, 255).to(torch.uint8)
    ori = _FILE_LUT[0].to(dev)[k8.long()]
    cf = _FILE_LUT[1].to(dev)[c8.long()]
    return ori, cf, k8, c8


def pmvo_maps_from_u8(k8, c8):
    """(ori, conf) fp32 planes from the two uint8 planes, as the loaders would produce them."""
    pmvo_maps_from_gabor(torch.zeros((1, 1), dtype=torch.int32, device=k8.device),
                         torch.zeros((1, 1), dtype=torch.float32, device=k8.device))     # builds the LUT
    return _FILE_LUT[0].to(k8.device)[k8.long()], _FILE_LUT[1].to(k8.device)[c8.long()]


def orientation_maps_device(images, device=None, gabor=None, return_codes=False):
    """The Gabor stage for a list of gray uint8 images, device-resident, views dealt to the ranks of an initialised
    torch.distributed group (the image-wide confidence maximum is per view, so views are independent) and
    all-gathered as 2 B/px uint8 planes.  Returns (ori [V,H,W,2], conf [V,H,W]) fp32 tensors on every rank, or with
    return_codes=True the 8-bit file codes themselves (best_ori [V,H,W], conf [V,H,W] uint8) for PMVO.from_u8."""
    from . import dist as mdist

    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    gabor = gabor or calOrientationGabor(device=device)
    V = len(images)
    H, W = (int(v) for v in images[0].shape)          # numpy arrays or device tensors (gabor.view takes both)
    local = []
    # views are independent: they rotate over two HIP streams, so that the DoG / finish launches of one view run in the tail of
    # the previous view's bank kernel (1.82 -> 1.79 ms per 1080p view, tools/bench_gabor.py --stage --streams 2)
    main = torch.cuda.current_stream(device)
    sts = gabor.__dict__.setdefault("_side_streams", None) or [torch.cuda.Stream(device=device) for _ in range(2)]
    gabor._side_streams = sts
    for st in sts:
        st.wait_stream(main)
    mine = [i for i in range(V) if mdist.owner(i) == mdist.rank()]
    for n, i in enumerate(mine):
        with torch.cuda.stream(sts[n % len(sts)]):
            _, _, _, k8, c8 = gabor.view(images[i])
            both = torch.stack([k8, c8], 0)
        both.record_stream(main)
        local.append(both)
    for st in sts:
        main.wait_stream(st)
    planes = mdist.all_gather_views(local, V, (2, H, W), torch.uint8, device)      # [V,2,H,W]
    if return_codes:
        return planes[:, 0].contiguous(), planes[:, 1].contiguous()
    ori, conf = pmvo_maps_from_u8(planes[:, 0], planes[:, 1])
    return ori, conf


def _orientation_arrays(image_u8, gabor, iter=1, threshold=0.0):
    """gray uint8 image -> (deg uint8 [H,W], conf uint8 [H,W], viz uint8 [H,W,3]) as calculate_orientation writes
    them; DoG and Gabor bank on the device (the device DoG reproduces scipy's float64 result bit for bit)."""
    dog = difference_of_gaussians_device(image_u8, 0.4, 10, gabor.device, out32=True)
    two, best, confidence = gabor(dog[None, None], None, iter, threshold=threshold)
    deg = torch.round(best[0, 0] / math.pi * 180).clamp(0, 255).to(torch.uint8).cpu().numpy()
    c8 = (confidence[0, 0] * 255 + 0.5).clamp(0, 255).to(torch.uint8).cpu().numpy()
    # RGB = (1, sin, cos) as cv2 BGR[::-1]; (x+1)/2 in float32, then x255 and the rounding in float64 like the host
    # formula it replaces, on the device
    ori = ((two[0].permute(1, 2, 0) + 1) / 2).double()
    viz = torch.cat([torch.ones_like(ori[..., :1]), ori], dim=2) * 255
    return deg, c8, torch.round(viz).clamp(0, 255).to(torch.uint8).cpu().numpy()


def _save_orientation_files(save_root, filename, deg, c8, viz):
    from PIL import Image

    # JPEG at quality 100 like the reference's cv2.imwrite; PNG with zlib level 1 (OpenCV's default level -- the
    # pixels are the same at any level, the encode is ~3x faster than PIL's default 6)
    low = filename.lower()
    kw = dict(quality=100) if low.endswith((".jpg", ".jpeg")) else (dict(compress_level=1) if low.endswith(".png") else {})
    Image.fromarray(deg).save(os.path.join(save_root, "best_ori", filename), **kw)
    Image.fromarray(np.repeat(c8[..., None], 3, axis=2)).save(os.path.join(save_root, "conf", filename),
                                                              **({} if "quality" in kw else kw))
    Image.fromarray(viz).save(os.path.join(save_root, "Ori", filename), **kw)


def calculate_orientation(image_dir, label_dir, save_root, filename=None, iter=1, threshold=0.0, gabor=None):
    """GaborFilter.py:164-224: gray image -> DoG -> Gabor bank -> best_ori/<file> (uint8 degrees),
    conf/<file> (uint8, x255+0.5, 3 channels), Ori/<file> (colour visualisation)."""
    from PIL import Image

    for sub in ("Ori", "conf", "best_ori"):
        os.makedirs(os.path.join(save_root, sub), exist_ok=True)
    gabor = gabor or calOrientationGabor()
    image = np.array(Image.open(image_dir).convert("L"))
    deg, c8, viz = _orientation_arrays(image, gabor, iter, threshold)
    _save_orientation_files(save_root, filename, deg, c8, viz)
    return deg, c8


def batch_generate(root, image_folder, io_threads=None):
    """GaborFilter.py:231-237.  With torch.distributed initialised the views are dealt to the ranks (the global
    maximum in the confidence is per image, so views are independent).  Image decoding and the three encodes per
    view (what the stage spends its time on once the filter takes 2 ms) run on a thread pool around the GPU work."""
    from concurrent.futures import ThreadPoolExecutor

    from PIL import Image

    from . import dist as mdist

    files = sorted(os.listdir(os.path.join(root, image_folder)))
    mine = [f for i, f in enumerate(files) if mdist.owner(i) == mdist.rank()]
    for sub in ("Ori", "conf", "best_ori"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    gabor = calOrientationGabor()

    def load(f):
        return np.array(Image.open(os.path.join(root, image_folder, f)).convert("L"))

    io_threads = min(16, os.cpu_count() or 1) if io_threads is None else io_threads
    with ThreadPoolExecutor(max(1, io_threads)) as pool:
        pending = []
        for f, image in zip(mine, pool.map(load, mine)):          # decoded ahead of the GPU, in order
            deg, c8, viz = _orientation_arrays(image, gabor)
            pending.append(pool.submit(_save_orientation_files, root, f, deg, c8, viz))
        for p in pending:
            p.result()
    mdist.barrier()
