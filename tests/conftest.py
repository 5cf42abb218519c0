import ast
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def fake_rccl_lib():
    """tests/lib/libfake_rccl.so, (re)built from tests/fake_rccl.cpp when missing or older than its source: the test-only
    stand-in for the RCCL entry points (MH_RCCL_LIB) that lets several ranks share one GPU."""
    import subprocess

    src = os.path.join(ROOT, "tests", "fake_rccl.cpp")
    out = os.path.join(ROOT, "tests", "lib", "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", src, "-o", out, "-lrt"])
    return out


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    if "scene_depth" in z.files:       # the fixture carries its own maps (see tools/gen_golden.py)
        meta["_scene"] = {k: z["scene_" + k] for k in ("depth", "ori", "conf", "mask")}
    return meta, z


_scene_cache = {}


def golden_scene(meta):
    """Regenerate the synthetic scene of a golden case (bit-identical; checked by checksum in the tests)."""
    from monohair_amd import synth

    key = (meta["V"], meta["H"], meta["W"], meta["seed"], meta["scale"], meta["rings"], meta["quantize"])
    if key not in _scene_cache and "_scene" in meta:
        import torch

        scene = {k: torch.from_numpy(np.array(v)) for k, v in meta["_scene"].items()}
        scene["cams"] = synth.make_cameras(meta["V"], meta["H"], meta["W"], scale=meta["scale"], rings=meta["rings"])
        scene["image_size"] = [meta["H"], meta["W"]]
        _scene_cache[key] = scene
    if key not in _scene_cache:
        _scene_cache[key] = synth.make_scene(meta["V"], meta["H"], meta["W"], seed=meta["seed"], scale=meta["scale"],
                                             rings=meta["rings"], quantize=meta["quantize"])
    return _scene_cache[key]


def golden_records(z):
    """[V,48] camera records assembled from the REFERENCE's own camera tensors stored in a golden file
    (pose, proj and torch.linalg.inv(pose[:3,:3]) as evaluated where the goldens were generated: MKL's
    3x3 inverse is not bit-reproducible across host CPUs, and the goldens must be compared like for like)."""
    V = z["cam_pose"].shape[0]
    rec = np.zeros((V, 48), np.float32)
    rec[:, 0:16] = z["cam_pose"].reshape(V, 16)
    rec[:, 16:32] = z["cam_proj"].reshape(V, 16)
    rec[:, 32:41] = z["cam_rinv"].reshape(V, 9)
    return rec


def scene_views(scene, records=None):
    """oracle.Views of a synth scene (host planes + camera records)."""
    import oracle
    from monohair_amd.camera import camera_records, cameras_from_list

    if records is None:
        records = camera_records(cameras_from_list(scene["cams"]))
    return oracle.Views(records, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(),
                        scene["conf"].cpu().numpy(), scene["mask"].cpu().numpy())


@pytest.fixture(scope="session")
def depth_offsets():
    return np.load(os.path.join(GOLDEN, "depth_offsets.npy"))
