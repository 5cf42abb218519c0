"""ctypes binding of libmhpmvo.so (include/mh_pmvo.h).  There is NO fallback: if the HIP library is
missing or a call fails, the product path raises -- it never routes through a CPU implementation."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmhpmvo.so")
_lib = None

vp = ctypes.c_void_p
ci = ctypes.c_int
cf = ctypes.c_float
cll = ctypes.c_longlong
csz = ctypes.c_size_t

_SIGS = {
    "mh_last_error": (ctypes.c_char_p, []),
    "mh_version": (ci, []),
    "mh_ctx_create": (ci, [ci, ctypes.POINTER(vp)]),
    "mh_ctx_destroy": (None, [vp]),
    "mh_ctx_alloc_views": (ci, [vp, ci, ci, ci]),
    "mh_ctx_set_view": (ci, [vp, ci, vp, vp, ci, vp, vp, vp, ci, vp]),
    "mh_ctx_set_view_u8": (ci, [vp, ci, vp, vp, ci, vp, vp, vp, vp, vp]),
    "mh_ctx_set_depth_offsets": (ci, [vp, vp, ci]),
    "mh_upload_async": (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
    "mh_upload_pinned": (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
    "mh_ctx_set_option": (ci, [vp, ctypes.c_char_p, ci]),
    "mh_ctx_set_lab_option": (ci, [vp, ctypes.c_char_p, ci]),
    "mh_debug_key_stats": (ci, [vp, ci]),
    "mh_project_gather": (ci, [vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mh_topk_views": (ci, [vp, vp, vp, ci, vp, vp, vp]),
    "mh_search_scratch_bytes": (csz, [vp, ci, ci]),
    "mh_search_counts_offset": (csz, [vp, ci, ci]),
    "mh_search_forward": (ci, [vp, vp, ci, ci, cf, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, csz, vp, vp, vp, vp, vp,
                               vp, vp]),
    "mh_forward_prepare": (ci, [vp, vp, ci, ci, cf, vp, vp, vp, vp, vp, csz, vp]),
    "mh_search_prepared": (ci, [vp, vp, ci, ci, cf, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mh_forward": (ci, [vp, vp, ci, ci, cf, ci, ci, vp, vp, vp, vp, vp, csz, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "mh_refine_loss": (ci, [vp, vp, vp, cf, cf, ci, ci, cf, vp, vp, vp, vp, vp, vp]),
    "mh_filter_points": (ci, [vp, vp, ci, ci, cf, cf, vp, vp, vp, vp, ci, cll, cll, vp]),
    "mh_filter_points_ordered": (ci, [vp, vp, ci, ci, cf, cf, vp, vp, vp, vp, ci, cll, cll, vp, vp]),
    "mh_project_points": (ci, [vp, ci, vp, ci, vp, vp, vp, vp, vp]),
    "mh_gather_pixels": (ci, [vp, ci, vp, ci, ci, vp, vp, vp]),
    "mh_compute_visible": (ci, [vp, vp, vp, csz, vp, vp]),
    "mh_sample_next": (ci, [vp, vp, vp, vp, vp, ci, ci, vp, vp]),
    "mh_reproject_ori": (ci, [vp, vp, vp, ci, ci, vp, vp]),
    "mh_prj_loss": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, cf, vp, vp, vp, vp, vp]),
    "mh_medoid_indexed": (ci, [vp, vp, vp, ci, ci, vp, vp, vp]),
    "mh_refine_loss_maps": (ci, [vp, vp, vp, cf, cf, ci, ci, cf, vp, vp, ci, cll, cll, vp]),
    "mh_refine_combine": (ci, [vp, vp, vp, vp, vp, cf, vp, vp, ci, vp]),
    "mh_medoid_dense": (ci, [vp, vp, ci, ci, vp, vp, vp]),
    "mh_medoid_segmented": (ci, [vp, vp, vp, ci, ci, vp, vp, vp]),
    "mh_replace_dissimilar": (ci, [vp, vp, vp, cf, ci, vp]),
    "mh_volume_pack": (ci, [vp, vp, vp, ci, ci, ci, vp, vp]),
    "mh_trace_seeds": (ci, [vp, vp, ci, ci, ci, vp, ci, cf, vp, vp, vp, vp]),
    "mh_trace_scalp": (ci, [vp, vp, ci, ci, ci, vp, vp, ci, cf, vp, vp, vp]),
    "mh_strands_accept": (ci, [ci, ci, ci, vp, vp, vp, vp, ci, vp, ci, ci, vp]),
    "mh_strands_compact": (ci, [vp, vp, vp, vp, vp, ci, ci, vp, vp]),
    "mh_knn_grid": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp]),
    "mh_nearest_distance": (ci, [vp, vp, ci, vp, ci, vp, ctypes.c_double, ctypes.c_double, vp, vp]),
    "mh_points_bbox": (ci, [vp, vp, ci, vp, vp]),
    "mh_grid_scratch_bytes": (csz, [ci]),
    "mh_grid_build": (ci, [vp, vp, vp, vp, ci, vp, csz, vp, vp, vp, vp, vp]),
    "mh_sort_scratch_bytes": (csz, [ci]),
    "mh_sort_keys": (ci, [vp, vp, ci, ci, vp, csz, vp, vp, vp]),
    "mh_voxel_group_scratch_bytes": (csz, [ci]),
    "mh_voxel_group": (ci, [vp, vp, ci, vp, ci, vp, ctypes.c_double, vp, vp, csz, vp, vp, vp, vp]),
    "mh_select_scratch_bytes": (csz, [ci]),
    "mh_select_rows": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, csz, vp]),
    "mh_segment_heads": (ci, [vp, vp, ci, vp, vp, vp, vp, csz, vp]),
    "mh_flag_less": (ci, [vp, vp, cf, ci, vp, vp]),
    "mh_buffers_differ": (ci, [vp, vp, vp, csz, vp, vp]),
    "mh_render_scratch_bytes": (csz, [ci, ci, ci, ci]),
    "mh_render_depth": (ci, [vp, vp, vp, ci, vp, ci, ci, ci, cf, vp, csz, vp, ci, vp]),
    "mh_comm_unique_id": (ci, [vp]),
    "mh_comm_init": (ci, [vp, vp, ci, ci, ctypes.POINTER(vp)]),
    "mh_comm_destroy": (ci, [vp]),
    "mh_volume_reduce": (ci, [vp, vp, ci, ci, ci, vp, ci, ci, ci, ci, vp, ci, vp]),
    "mh_volume_gather": (ci, [vp, vp, ci, ci, ci, vp, vp, ci, ci, ci, ci, vp, vp]),
    "mh_render_strands_scratch_bytes": (csz, [ci, ci, ci, ci, ci]),
    "mh_render_strands": (ci, [vp, vp, vp, ci, vp, ci, vp, vp, ci, ci, ci, cf, ci, ci, ci, cf, vp, csz, vp, vp]),
    "mh_mat_write_sparse": (ci, [ctypes.c_char_p, vp, csz, csz, vp, vp, csz, ci]),
    "mh_mat_sparse_open": (ci, [ctypes.c_char_p, vp, csz, csz, ctypes.POINTER(vp)]),
    "mh_mat_sparse_touch": (ci, [vp, vp, csz]),
    "mh_mat_sparse_store": (ci, [vp, vp, vp, csz]),
    "mh_mat_sparse_store_voxels": (ci, [vp, vp, vp, ci, csz, ci, ci, ci]),
    "mh_mat_sparse_close": (ci, [vp]),
    "mh_gabor_bank": (ci, [vp, vp, ci, ci, vp, vp, vp, vp]),
    "mh_gabor_set_bank": (ci, [vp, vp]),
    "mh_dog_scratch_bytes": (csz, [ci, ci]),
    "mh_dog": (ci, [vp, vp, ci, ci, ci, vp, ci, vp, ci, vp, vp, vp, vp]),
    "mh_gabor_view_scratch_bytes": (csz, [ci, ci]),
    "mh_gabor_view": (ci, [vp, vp, ci, ci, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp]),
}

EXPORTS = sorted(_SIGS)


class MhError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (monohair_amd/lib/libmhpmvo.so)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MhError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mh_last_error()
        raise MhError("%s failed (%d): %s" % (what or "libmhpmvo", rc, msg.decode() if msg else "?"))


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
