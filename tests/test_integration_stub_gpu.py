"""GPU: the raw ctypes binding shown in INTEGRATION.md §3 (no monohair_amd Python on the call path) drives the C ABI
and gives what the PMVO class gives."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_ctypes_stub_of_integration_md():
    from monohair_amd import synth
    from monohair_amd.camera import camera_records, cameras_from_list
    from monohair_amd.pmvo import PMVO

    V, H, W, patch, N = 24, 120, 80, 5, 333
    scene = synth.make_scene(V, H, W, seed=4)
    cams = cameras_from_list(scene["cams"])
    recs = camera_records(cams)
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)

    # ---- the stub -------------------------------------------------------------------------
    L = ctypes.CDLL(os.path.join(ROOT, "monohair_amd", "lib", "libmhpmvo.so"))
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.mh_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.mh_ctx_alloc_views.argtypes = [vp, ci, ci, ci]
    L.mh_ctx_set_view.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp, ci, vp]
    L.mh_project_gather.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    L.mh_ctx_destroy.argtypes = [vp]
    L.mh_last_error.restype = ctypes.c_char_p
    ctx = vp()
    assert L.mh_ctx_create(0, ctypes.byref(ctx)) == 0
    assert L.mh_ctx_alloc_views(ctx, V, H, W) == 0
    stream = torch.cuda.current_stream().cuda_stream
    for i, k in enumerate(cams):
        d, o, c, m = (torch.from_numpy(np.ascontiguousarray(a[k])).cuda().float() for a in (depths, Ori, Conf, masks))
        rec = np.ascontiguousarray(recs[i], dtype=np.float32)
        rc = L.mh_ctx_set_view(ctx, i, rec.ctypes.data, d.data_ptr(), 3, o.data_ptr(), c.data_ptr(), m.data_ptr(), 3,
                               stream)
        assert rc == 0, L.mh_last_error()
        torch.cuda.synchronize()
    pts = torch.from_numpy(synth.candidate_points(res=32, seed=1)[:N]).float().cuda()
    f = dict(dtype=torch.float32, device="cuda")
    P = patch * patch
    vis, ori, conf, mask = (torch.empty((V, N), **f), torch.empty((V, N, 2), **f), torch.empty((V, N), **f),
                            torch.empty((V, N), **f))
    ori_patch, conf_patch, pixf = torch.empty((V, N, P, 2), **f), torch.empty((V, N, P), **f), torch.empty((V, N, 2), **f)
    rc = L.mh_project_gather(ctx, pts.data_ptr(), N, patch, vis.data_ptr(), ori.data_ptr(), conf.data_ptr(),
                             mask.data_ptr(), ori_patch.data_ptr(), conf_patch.data_ptr(), pixf.data_ptr(), stream)
    assert rc == 0, L.mh_last_error()
    torch.cuda.synchronize()
    # bad arguments come back as a status + message, not a crash
    assert L.mh_project_gather(ctx, None, N, patch, None, None, None, None, None, None, None, stream) != 0
    assert b"mh_project_gather" in L.mh_last_error()
    L.mh_ctx_destroy(ctx)
    # -----------------------------------------------------------------------------------------------------------

    pm = PMVO(cams, depths, Ori, Conf, masks, device="cuda:0", image_size=[H, W], patch_size=patch,
              visible_threshold=1, conf_threshold=0.15)
    pm.Compute_Visible_and_Ori(pts.cpu().numpy())
    assert torch.equal(vis, pm.visible) and torch.equal(ori, pm.Ori) and torch.equal(conf, pm.Conf)
    assert torch.equal(mask, pm.mask) and torch.equal(ori_patch, pm.Ori_patch) and torch.equal(conf_patch, pm.Conf_patch)
