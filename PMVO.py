#!/usr/bin/env python
"""PMVO.py -- drop-in entry point of the exterior-geometry optimisation, same command line, YAML keys and
files as the reference's PMVO.py (__main__ :805-880, config_parser :767-800):

    python PMVO.py --yaml=configs/reconstruct/<case> [--PMVO.infer_inner] [--PMVO.optimize=] [--gpu=N] [--a.b=v]
    python -m torch.distributed.run --nproc-per-node 8 PMVO.py --yaml=...        (one process per MI355X)

in : data/<case>/ours/cam_params.json, capture_images/, render_depth/<view>.npy, best_ori/, conf/, hair_mask/,
     ours/colmap_points.obj, ours/bust_long_tsfm.obj, ours/scalp_tsfm.obj [, ours/raw.npy]
out: data/<case>/output/<name>/options.yaml, optimize/*.npy, refine/*.npy, refine|full/{Ori3D,Occ3D}.mat

All arithmetic runs in the HIP library (monohair_amd/lib/libmhpmvo.so); this file is host orchestration.
"""
import os
import sys

import numpy as np

from monohair_amd import options
from monohair_amd.camera import load_cam, parsing_camera
from monohair_amd.pmvo import PMVO, filter_negative_points, optimize, refine  # noqa: F401  (re-exported)
from monohair_amd.timing import stage
from monohair_amd.pmvo_utils import (Load_Ori_And_Conf, load_bust, load_colmap_points, load_depth,  # noqa: F401
                                     load_depth_plane, load_maps_u8, load_mask, read_obj)


def config_parser(argv=None):
    print("Process ID: {}".format(os.getpid()))
    opt_cmd = options.parse_arguments(sys.argv[1:] if argv is None else argv)
    args = options.set(opt_cmd=opt_cmd)
    args.output_path = os.path.join(args.data.root, args.data.case, args.output_root, args.name)
    from monohair_amd import dist as mdist

    if mdist.rank() == 0:       # one writer: another rank could otherwise read a half-written options.yaml
        os.makedirs(args.output_path, exist_ok=True)
        options.save_options_file(args)
    mdist.barrier()
    args.data.root = os.path.join(args.data.root, args.data.case)
    args.bbox_min = np.array(args.bbox_min)
    args.bust_to_origin = np.array(args.bust_to_origin)
    for key in ("strands_path", "bust_path", "raw_points_path", "depth_path", "Ori2D_path", "Conf_path", "mask_path"):
        args.data[key] = os.path.join(args.data.root, args.data[key])
    args.image_camera_path = os.path.join(args.data.root, args.image_camera_path)
    args.save_root = os.path.join(args.output_path, "optimize")
    if args.PMVO.infer_inner and not args.PMVO.optimize:
        args.save_path = os.path.join(args.output_path, "full")
    else:
        args.save_path = os.path.join(args.output_path, "refine")
    if mdist.rank() == 0:
        os.makedirs(args.save_path, exist_ok=True)
    mdist.barrier()
    return args


def load_views(camera, args):
    """The map upload of the reference's __main__ (PMVO.py:822-840).  The 8-bit files are uploaded as pixel codes
    and decoded on the GPU (PMVO.from_u8: same values as Load_Ori_And_Conf/load_mask + the float constructor);
    with `data.maps_pack` set, one memory-mapped pack replaces the four files per view (written on first use)."""
    from monohair_amd import dist as mdist
    from monohair_amd import mapspack

    kw = dict(device=args.device, image_size=args.data.image_size, patch_size=args.PMVO.patch_size,
              visible_threshold=args.PMVO.visible_threshold, conf_threshold=args.PMVO.conf_threshold)
    pack = args.data.get("maps_pack")
    if pack:
        pack = pack if os.path.isabs(pack) else os.path.join(args.data.root, pack)
        # rank 0 alone decides whether the pack has to be written; EVERY rank then takes the same barrier (a rank
        # that tested for the file itself could see rank 0's finished pack, skip the barrier and pair its next
        # collective with the others' barrier)
        if mdist.rank() == 0 and not os.path.exists(pack):
            print("writing maps pack", pack)
            mapspack.pack_case(camera, args.data.Ori2D_path, args.data.Conf_path, args.data.mask_path,
                               args.data.depth_path, pack)
        mdist.barrier()
        m = mapspack.read_pack(pack, views=list(camera.keys()))
        return PMVO.from_u8(camera, m["depth"], m["ori"], m["conf"], m["mask"], **kw)
    ori, conf, mask = load_maps_u8(camera, args.data.Ori2D_path, args.data.Conf_path, args.data.mask_path)
    depths = load_depth_plane(camera, args.data.depth_path)
    return PMVO.from_u8(camera, depths, ori, conf, mask, **kw)


def apply_reference_host(pmvo, spec):
    """--PMVO.reference_host=<file.json | inline JSON>: the rounding facts of the host the reference's CPU run is compared on
    (`python tools/probe_mkl_forms.py --emit-options` there): reproject_fma_min_cols -- the column count from which MKL's
    sgemm in Camera.reprojection (Utils/Camera_utils.py:81-106) switches to its threaded fma-chain kernel, which moves with
    the thread count (8 threads: 28 445, the default; 4: 14 223; 2: 21 334; 1: never) -- and sum_block.  Pinned end to end
    at 1 / 2 / 4 / 8 threads by tests/golden/pmvo_threads.npz."""
    if not spec:
        return
    import json

    text = str(spec)
    host = json.loads(open(text).read() if os.path.exists(text) else text)
    for key in ("reproject_fma_min_cols", "sum_block", "reproject_rule"):
        if key in host:
            pmvo.set_option(key, int(host[key]))
    print("reference host: %s" % {k: host[k] for k in host if k in ("threads", "reproject_fma_min_cols", "sum_block")})


def main(argv=None):
    from scipy.spatial import KDTree

    from monohair_amd import dist as mdist

    print("Run PMVO...")
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as tdist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MH_DIST_BACKEND=gloo + MH_DEVICE_OVERRIDE: test hooks to run several ranks on one GPU (RCCL refuses that)
        if os.environ.get("MH_DIST_BACKEND", "nccl") == "nccl":
            tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
        else:
            tdist.init_process_group(backend=os.environ["MH_DIST_BACKEND"])
    args = config_parser(argv)

    vertices, faces, normals = load_bust(args.data.bust_path)
    vertices += args.bust_to_origin
    bust_tree = KDTree(data=vertices)
    scalp_vertices, _ = read_obj(os.path.join(args.data.root, "ours/scalp_tsfm.obj"))
    scalp_vertices += args.bust_to_origin
    scalp_tree = KDTree(data=scalp_vertices)
    scalp_max = np.max(scalp_vertices, axis=0)

    camera = parsing_camera(load_cam(args.image_camera_path), os.path.join(args.data.root, "capture_images"))
    print("num of view:", len(camera))
    with stage("load maps -> device", args.device):
        pmvo = load_views(camera, args)
    pmvo.set_head(bust_tree, scalp_tree, scalp_max)
    apply_reference_host(pmvo, args.PMVO.get("reference_host"))

    if args.PMVO.optimize:
        print("load raw mesh...")
        with stage("candidate points"):
            points = load_colmap_points(args.data.raw_points_path, args.bbox_min, args.bust_to_origin, 0.005 / 4,
                                        [512, 512, 384], True, args.PMVO.num_sample_per_grid)
        raw_points = points.copy()
        print("total points:", points.shape[0])
        print("filter low conf points...")
        if args.PMVO.filter_point:
            with stage("filter_negative_points", args.device):
                surface_index, surface_points, filter_index = filter_negative_points(points, pmvo, args)
            points = surface_points
            writer = None
            if mdist.rank() == 0:
                # surface.npy / filter_unvisible.npy (PMVO.py:853-854: float64 rows of the raw candidates, 7 + 1.5 MB at the
                # headline size) are selected and written by a worker while optimize() iterates; joined before they are read
                import threading

                os.makedirs(args.save_root, exist_ok=True)
                failed = []

                def _save_masks():
                    try:
                        np.save(os.path.join(args.save_root, "surface.npy"), raw_points[:len(surface_index)][surface_index])
                        np.save(os.path.join(args.save_root, "filter_unvisible.npy"),
                                raw_points[:len(filter_index)][filter_index])
                    except BaseException as e:
                        failed.append(e)

                writer = threading.Thread(target=_save_masks)
                writer.start()
        print("process points:", points.shape[0])
        with stage("optimize", args.device):
            optimize(points, pmvo, args)
        if args.PMVO.filter_point:
            if writer is not None:
                writer.join()
                if failed:
                    raise failed[0]
            mdist.barrier()
        select_points = np.load(args.save_root + "/select_p.npy")
        select_ori = np.load(args.save_root + "/select_o.npy")
        min_loss = np.load(args.save_root + "/min_loss.npy")
        filter_unvisible_points = np.load(args.save_root + "/filter_unvisible.npy")
        refine(select_points, select_ori, min_loss, pmvo, filter_unvisible_points, args, infer_inner=False,
               threshold=args.PMVO.threshold, genrate_ori_only=False, return_dense=False)
    else:
        select_points = np.load(args.save_root + "/select_p.npy")
        select_ori = np.load(args.save_root + "/select_o.npy")
        min_loss = np.load(args.save_root + "/min_loss.npy")
        filter_unvisible_points = np.load(args.save_root + "/filter_unvisible.npy")
        refine(select_points, select_ori, min_loss, pmvo, filter_unvisible_points, args,
               infer_inner=args.PMVO.infer_inner, threshold=args.PMVO.threshold, genrate_ori_only=True,
               return_dense=False)


if __name__ == "__main__":
    main()
