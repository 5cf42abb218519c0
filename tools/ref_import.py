"""Import the MonoHair reference (read-only, /root/reference) in THIS container only.

Used by tools/gen_golden.py to produce the golden vectors under tests/golden/.
Nothing here travels in any form that contains reference code: the reference is
imported from where it lies, with empty stand-in modules for the third-party
packages that are not installed (they are never *called* on the paths we run --
only imported at module top level).  SURVEY.md Appendix B is the recipe.
"""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


PINNED_TORCH = "2.10."        # the goldens under tests/golden/ are outputs of CPU torch of this version


def import_reference(gabor=False):
    import os

    import torch

    print("gen_golden: CPU torch %s" % torch.__version__)
    if not torch.__version__.startswith(PINNED_TORCH) and not os.environ.get("MH_GOLDEN_ANY_TORCH"):
        raise SystemExit("the golden vectors are pinned to CPU torch %sx: torch.topk's tie order and ATen's summation order "
                         "are part of what they record (DESIGN.md §5).  This is torch %s; set MH_GOLDEN_ANY_TORCH=1 to "
                         "regenerate them deliberately with another version." % (PINNED_TORCH, torch.__version__))

    sys.dont_write_bytecode = True
    for name in ("cv2", "trimesh", "open3d", "termcolor"):
        _stub(name)
    _stub("easydict", EasyDict=dict)
    if gabor:
        _stub("imageio")
        sk = _stub("skimage")
        skf = _stub("skimage.filters", difference_of_gaussians=lambda *a, **k: None)
        sk.filters = skf
        tv = _stub("torchvision")
        tvt = _stub("torchvision.transforms")
        tvu = _stub("torchvision.utils", save_image=lambda *a, **k: None)
        tv.transforms, tv.utils = tvt, tvu
        ident = lambda self, *a, **k: self
        torch.Tensor.cuda = ident
        torch.nn.Module.cuda = ident
    # extra stand-ins needed only to IMPORT HairGrow.py / Utils/Utils.py (never called on the paths we run)
    o3 = sys.modules["open3d"]
    o3.core = _stub("open3d.core")
    tm = sys.modules["trimesh"]
    tmv = _stub("trimesh.visual", texture=None, TextureVisuals=None)
    tm.visual = tmv
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import PMVO as ref_pmvo  # noqa
    import Utils.Camera_utils as ref_cam  # noqa
    import Utils.PMVO_utils as ref_utils  # noqa

    out = dict(PMVO=ref_pmvo, Camera_utils=ref_cam, PMVO_utils=ref_utils)
    if gabor:
        import preprocess_capture_data.GaborFilter as ref_gabor  # noqa

        out["GaborFilter"] = ref_gabor
    return out
