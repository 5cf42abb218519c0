"""GPU: strand tracing (csrc/hairgrow.hip + monohair_amd.hairgrow) against the CPU oracle and the reference's own
HairGrowing run (tests/golden/hairgrow.npz)."""
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def setup():
    from monohair_amd.hairgrow import HairGrowing

    z = np.load(os.path.join(GOLDEN, "hairgrow.npz"))
    occ = z["occ"].transpose(2, 1, 0)[..., None]           # [Z,Y,X,1] as get_ground_truth_3D_occ returns
    ori = z["ori"].transpose(2, 1, 0, 3)                    # [Z,Y,X,3]
    hg = HairGrowing(None, None, device=DEV, occ=occ, ori=ori)
    vol = oracle.Volume(occ[..., 0], ori)
    return z, hg, vol


def test_kernels_match_oracle_exactly(setup):
    z, hg, vol = setup
    assert np.array_equal(hg._vox.cpu().numpy(), vol.vox)
    rng = np.random.default_rng(0)
    seeds = (np.argwhere(vol.vox[..., 3] != 0)[:, ::-1] + rng.random((int((vol.vox[..., 3] != 0).sum()), 3))).astype(
        np.float32)
    out, first, ln = hg._trace_seeds(torch.from_numpy(seeds).to(DEV), 0.8)
    o_out, o_first, o_ln = oracle.trace_seeds(vol, seeds, 0.8)
    assert np.array_equal(first.cpu().numpy(), o_first) and np.array_equal(ln.cpu().numpy(), o_ln)
    out = out.cpu().numpy()
    for i in range(len(seeds)):
        assert np.array_equal(out[i, o_first[i]:o_first[i] + o_ln[i]], o_out[i, o_first[i]:o_first[i] + o_ln[i]])
    sp, sl = hg._trace_scalp(torch.from_numpy(z["scalp_points"]).to(DEV), torch.from_numpy(z["scalp_normals"]).to(DEV),
                             0.8)
    o_sp, o_sl = oracle.trace_scalp(vol, z["scalp_points"], z["scalp_normals"], 0.8)
    assert np.array_equal(sl.cpu().numpy(), o_sl)
    sp = sp.cpu().numpy()
    for i in range(len(o_sl)):
        assert np.array_equal(sp[i, :o_sl[i]], o_sp[i, :o_sl[i]])


def test_guide_strands_match_reference(setup):
    z, hg, vol = setup
    torch.manual_seed(77)
    strands, num_root = hg.GenerateGuideStrandFromScalp(torch.from_numpy(z["scalp_points"].copy()),
                                                        torch.from_numpy(z["scalp_normals"].copy()), None, 0.8)
    assert num_root == int(z["guide_num_root"])
    assert np.array_equal(np.array([s.shape[0] for s in strands]), z["guide_len"])
    assert np.array_equal(torch.cat(strands).cpu().numpy(), z["guide_pts"])


def test_random_segments_match_reference_and_hair_file(setup, tmp_path):
    from monohair_amd.pmvo_utils import load_strand, save_hair_strands

    z, hg, vol = setup
    torch.manual_seed(77)
    strands = hg.randomlyGenerateSegments(0.8)
    assert np.array_equal(np.array([s.shape[0] for s in strands]), z["random_len"])
    assert np.array_equal(torch.cat(strands).cpu().numpy(), z["random_pts"])
    world = hg.VoxelToWorld(strands, np.zeros(3, np.float32))
    p = str(tmp_path / "seg.hair")
    save_hair_strands(p, world, np.zeros(3), translate=False)
    seg, pts = load_strand(p)
    assert seg == [int(x) for x in z["random_len"]]
    assert np.allclose(pts, np.concatenate(world))


def test_per_seed_methods_equal_the_batched_drivers(setup):
    """trace / traceFromScalp (one seed per call, the reference's signatures) give the strands of the batched kernels"""
    z, hg, vol = setup
    W, H, Z = hg.W, hg.H, hg.Z
    sp, sn = torch.from_numpy(z["scalp_points"]), torch.from_numpy(z["scalp_normals"])
    o_sp, o_sl = oracle.trace_scalp(vol, z["scalp_points"], z["scalp_normals"], 0.8)
    for i in range(0, len(sp), max(1, len(sp) // 25)):
        s = hg.traceFromScalp(sp[i].clone(), sn[i].clone(), 0.8, W, H, Z, None)
        if o_sl[i] <= 0:
            assert s is None
        else:
            assert np.array_equal(s.cpu().numpy(), o_sp[i, :o_sl[i]])
    nz = np.argwhere(vol.vox[..., 3] != 0)[:, ::-1].astype(np.float32)
    flag = np.zeros((Z, H, W), np.float32)
    torch.manual_seed(5)
    jit = torch.rand(len(nz[:40]), 3)
    torch.manual_seed(5)
    for i in range(40):
        seed = torch.from_numpy(nz[i].copy())
        want_seed = (nz[i] + 0.5 + jit[i].numpy() * 0.5).astype(np.float32)
        got = hg.trace(seed, flag, 0.8, W, H, Z)
        assert np.array_equal(seed.numpy(), want_seed)              # shifted in place, like the reference
        o_out, o_first, o_ln = oracle.trace_seeds(vol, want_seed[None], 0.8)
        if o_ln[0] >= 5:
            assert np.array_equal(got.cpu().numpy(), o_out[0, o_first[0]:o_first[0] + o_ln[0]])
        else:
            assert got is False
    flag[:] = 3
    assert hg.trace(torch.from_numpy(nz[0].copy()), flag, 0.8, W, H, Z) is False
