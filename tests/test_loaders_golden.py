"""SURVEY.md §8a row 19 pinned to the reference: tests/golden/loaders.npz holds what the reference's OWN
Load_Ori_And_Conf / load_mask / load_depth (Utils/PMVO_utils.py:255-313) returned for PNG / npy files whose
pixel codes are stored next to them (tools/gen_golden_r2.py loaders; view 000 contains every code 0..255,
orientation codes above 180 included -- those wrap modulo 256 in the reference's uint8 arithmetic)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN


def _write_tree(z, root):
    from PIL import Image

    views = [str(v) for v in z["views"]]
    for d in ("best_ori", "conf", "hair_mask", "render_depth"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    for v in views:
        Image.fromarray(z["in_ori_" + v], "L").save(os.path.join(root, "best_ori", v + ".png"))
        Image.fromarray(np.repeat(z["in_conf_" + v][..., None], 3, -1), "RGB").save(os.path.join(root, "conf", v + ".png"))
        Image.fromarray(np.ascontiguousarray(z["in_mask_bgr_" + v][..., ::-1]), "RGB").save(
            os.path.join(root, "hair_mask", v + ".png"))
        np.save(os.path.join(root, "render_depth", v + ".npy"), z["in_depth_" + v])
    return views


def test_float_loaders_equal_the_reference(tmp_path):
    from monohair_amd import pmvo_utils as U

    z = np.load(os.path.join(GOLDEN, "loaders.npz"))
    views = _write_tree(z, str(tmp_path))
    cam = {v: None for v in views}
    p = lambda d: os.path.join(str(tmp_path), d)   # noqa: E731
    Ori, Conf = U.Load_Ori_And_Conf(cam, p("best_ori"), p("conf"))
    mask = U.load_mask(cam, p("hair_mask"))
    depth = U.load_depth(cam, p("render_depth"))
    for v in views:
        assert Ori[v].dtype == z["ref_Ori_" + v].dtype and np.array_equal(Ori[v], z["ref_Ori_" + v]), v
        assert Conf[v].dtype == z["ref_Conf_" + v].dtype and np.array_equal(Conf[v], z["ref_Conf_" + v]), v
        assert mask[v].dtype == z["ref_mask_" + v].dtype and np.array_equal(mask[v], z["ref_mask_" + v]), v
        assert depth[v].dtype == z["ref_depth_" + v].dtype and np.array_equal(depth[v], z["ref_depth_" + v]), v


def test_code_table_equals_the_reference_for_every_pixel_code(tmp_path):
    """map_code_lut()[code] == float32(reference loader output) for all 256 codes of all three map kinds, and
    load_maps_u8 / load_depth_plane return exactly the codes / channel 0 the reference decoded."""
    from monohair_amd import pmvo_utils as U

    z = np.load(os.path.join(GOLDEN, "loaders.npz"))
    views = _write_tree(z, str(tmp_path))
    cam = {v: None for v in views}
    p = lambda d: os.path.join(str(tmp_path), d)   # noqa: E731
    o8, c8, m8 = U.load_maps_u8(cam, p("best_ori"), p("conf"), p("hair_mask"), threads=2)
    d0 = U.load_depth_plane(cam, p("render_depth"), threads=2)
    lut = U.map_code_lut()
    seen = [set(), set(), set()]
    for v in views:
        assert np.array_equal(o8[v], z["in_ori_" + v]) and np.array_equal(c8[v], z["in_conf_" + v])
        assert np.array_equal(m8[v], z["in_mask_bgr_" + v][..., 0])
        # the float32 cast is the one PMVO.__init__ applies to the loaders' float64 arrays (PMVO.py:23-26)
        assert np.array_equal(lut[o8[v]][..., 0:2], z["ref_Ori_" + v].astype(np.float32)), v
        assert np.array_equal(lut[c8[v]][..., 2], z["ref_Conf_" + v].astype(np.float32)), v
        assert np.array_equal(lut[m8[v]][..., 3], z["ref_mask_" + v][..., 0].astype(np.float32)), v
        assert np.array_equal(d0[v], z["ref_depth_" + v][..., 0]), v
        for s, a in zip(seen, (o8[v], c8[v], m8[v])):
            s.update(np.unique(a).tolist())
    assert all(len(s) == 256 for s in seen), "the fixture must exercise every pixel code"


@pytest.mark.gpu
def test_gpu_table_decode_equals_the_reference_float_maps(tmp_path):
    """mh_ctx_set_view_u8 (pack kernel + 256-entry table) must leave the same resident records as the float
    constructor fed with the REFERENCE's decoded maps: every pixel of every view is gathered back and compared."""
    import torch

    from monohair_amd import pmvo_utils as U
    from monohair_amd.camera import Camera
    from monohair_amd.pmvo import PMVO

    z = np.load(os.path.join(GOLDEN, "loaders.npz"))
    views = [str(v) for v in z["views"]]
    H, W = z["in_ori_" + views[0]].shape
    cams = {v: Camera([1.2, 1.2, 0.0, 0.0], np.eye(4), v) for v in views}
    a = PMVO(cams, {v: z["ref_depth_" + v] for v in views}, {v: z["ref_Ori_" + v] for v in views},
             {v: z["ref_Conf_" + v] for v in views}, {v: z["ref_mask_" + v] for v in views}, device="cuda:0",
             image_size=[H, W], patch_size=3, conf_threshold=0.15)
    b = PMVO.from_u8(cams, {v: z["in_depth_" + v] for v in views}, {v: z["in_ori_" + v] for v in views},
                     {v: z["in_conf_" + v] for v in views}, {v: np.ascontiguousarray(z["in_mask_bgr_" + v][..., 0]) for v in views},
                     device="cuda:0", image_size=[H, W], patch_size=3, conf_threshold=0.15)
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    uv = torch.from_numpy(np.stack([rr.ravel(), cc.ravel()], 1))
    for i, v in enumerate(views):
        ra, ma = a._gather(uv, i, 1, want_mask=True)
        rb, mb = b._gather(uv, i, 1, want_mask=True)
        assert torch.equal(ra, rb) and torch.equal(ma, mb), v
        ref = np.concatenate([z["ref_Ori_" + v].reshape(-1, 2), z["ref_Conf_" + v].reshape(-1, 1),
                              z["ref_depth_" + v][..., 0].reshape(-1, 1)], 1).astype(np.float32)
        assert np.array_equal(rb[:, 0].cpu().numpy(), ref), v
        assert np.array_equal(mb[:, 0].cpu().numpy(), z["ref_mask_" + v][..., 0].reshape(-1).astype(np.float32)), v
