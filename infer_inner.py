#!/usr/bin/env python
"""infer_inner.py -- the caller on the far side of the PMVO path, same command line and files as the reference's
infer_inner.py (:30-90):

    python infer_inner.py --yaml=configs/reconstruct/<case> [--infer_inner.render_data=] [--infer_inner.run_mvs=]

render_data stage (:41-73): strand segments are traced on the exterior volume refine/{Occ3D,Ori3D}.mat
(HairGrowing.randomlyGenerateSegments, csrc/hairgrow.hip), written to refine/render_segments.hair, and drawn over the bust
for every camera of `camera_path` into data/<case>/imgs/<view>/{bust_depth,undirectional_map,mask,hair_depth}.png
(monohair_amd.render.render_data, csrc/raster.hip: mh_render_strands) -- the inputs of DeepMVSHair.
run_mvs stage (:77-90): DeepMVSHair itself (a ViT whose weights are not part of the reference repository, SURVEY.md §2
row 15) is outside this package; when its output data/<case>/ours/raw.npy is present the second PMVO pass
(`PMVO.py --PMVO.infer_inner --PMVO.optimize=`, what the reference spawns with os.system) is run in-process.
"""
import os
import sys

import numpy as np

from monohair_amd import options


def get_config(argv=None):
    print("Process ID: {}".format(os.getpid()))
    opt_cmd = options.parse_arguments(sys.argv[1:] if argv is None else argv)
    args = options.set(opt_cmd=opt_cmd)
    args.output_path = os.path.join(args.data.root, args.data.case, args.output_root, args.name)
    args.data.root = os.path.join(args.data.root, args.data.case)
    return args


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = get_config(argv)
    root = args.data.root
    os.makedirs(os.path.join(root, "ours"), exist_ok=True)
    if args.infer_inner.render_data:
        from monohair_amd.camera import load_cam, parsing_camera
        from monohair_amd.hairgrow import HairGrowing
        from monohair_amd.pmvo_utils import load_bust, save_hair_strands
        from monohair_amd.render import render_data

        save_path = os.path.join(args.output_path, "refine")
        solver = HairGrowing(os.path.join(save_path, "Occ3D.mat"), os.path.join(save_path, "Ori3D.mat"), args.device,
                             args.data.image_size)
        strands = solver.randomlyGenerateSegments(args.HairGenerate.grow_threshold)
        strands = solver.VoxelToWorld(strands)
        save_hair_strands(os.path.join(save_path, "render_segments.hair"), strands, None, translate=False)
        vertices, faces, _ = load_bust(os.path.join(root, args.data.bust_path))
        vertices = vertices + np.array(args.bust_to_origin)
        image_path = os.path.join(root, "trainning_images/capture_images")
        camera = parsing_camera(load_cam(args.camera_path), image_path if os.path.isdir(image_path) else None)
        print("render %d segments over the bust, %d views" % (len(strands), len(camera)))
        render_data(camera, strands, vertices, faces, [1280, 720], os.path.join(root, "imgs"), device=args.device)
    if args.infer_inner.run_mvs:
        raw = os.path.join(root, "ours", "raw.npy")
        if not os.path.exists(raw):
            raise SystemExit("infer_inner: %s is missing -- it is the output of DeepMVSHair (submodules/DeepMVSHair of the "
                             "reference, network weights not included there); run it on data/<case>/imgs/ and rerun, or "
                             "pass --infer_inner.run_mvs= to stop after the render stage" % raw)
        import PMVO as pmvo_cli

        # the second PMVO pass sees every option of this command line (name / seed / output_root / bbox / bust_to_origin /
        # camera path ... all decide where it finds refine/select_p.npy and with what geometry) except this script's own
        # switches
        keep = [a for a in argv if not a.startswith("--infer_inner")]
        pmvo_cli.main(keep + ["--PMVO.infer_inner", "--PMVO.optimize="])


if __name__ == "__main__":
    main()
