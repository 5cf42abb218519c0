"""Pin the strand-tracing oracle (oracle/hairgrow_oracle.c) against the reference's HairGrowing run on a small
synthetic fitted volume (tools/gen_golden_more.py hairgrow).  CPU only."""
import os

import numpy as np
import torch

import oracle
from conftest import GOLDEN


def load():
    z = np.load(os.path.join(GOLDEN, "hairgrow.npz"))
    occ_xyz, ori_xyz = z["occ"], z["ori"]
    # the readers hand HairGrowing [Z,Y,X] arrays (PMVO_utils.py:86-113)
    vol = oracle.Volume(occ_xyz.transpose(2, 1, 0), ori_xyz.transpose(2, 1, 0, 3))
    assert np.array_equal(vol.vox[..., 3], z["vol_occ_zyx"])
    return z, vol


def jitter(n_calls, seed=77):
    torch.manual_seed(seed)
    return torch.rand(n_calls, 3).numpy()       # == n_calls consecutive torch.rand_like(seedPos) draws


def split(pts, lens):
    o = np.concatenate([[0], np.cumsum(lens)])
    return [pts[o[i]:o[i + 1]] for i in range(len(lens))]


def test_guide_strands_match_reference():
    z, vol = load()
    n_occ = int((vol.vox[..., 3] != 0).sum())
    strands, num_root, flag = oracle.generate_guide_strands(vol, z["scalp_points"], z["scalp_normals"], float(z["thr"]),
                                                            jitter(2 * n_occ))
    assert num_root == int(z["guide_num_root"])
    ref = split(z["guide_pts"], z["guide_len"])
    assert len(strands) == len(ref)
    assert np.array_equal(np.array([len(s) for s in strands]), z["guide_len"])
    assert np.array_equal(np.concatenate(strands), z["guide_pts"])
    assert max(len(s) for s in strands) > 25 and flag.max() >= 3      # long strands and the flag gate are exercised


def test_random_segments_match_reference():
    z, vol = load()
    n_occ = int((vol.vox[..., 3] != 0).sum())
    strands, flag = oracle.randomly_generate_segments(vol, float(z["thr"]), jitter(3 * n_occ))
    assert np.array_equal(np.array([len(s) for s in strands]), z["random_len"])
    assert np.array_equal(np.concatenate(strands), z["random_pts"])
