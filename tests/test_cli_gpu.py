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
