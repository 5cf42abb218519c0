"""Test helper (not a test): the view-sharded Gabor stage (SURVEY.md §8e "map distribution"; GaborFilter.py:231-237) run as N
ranks under torch.distributed.run (gloo ranks sharing the test GPU) or as one process.  Writes, into --out:
  codes_rank<r>.npz   what orientation_maps_device(images, return_codes=True) returned on rank r (best_ori, conf codes [V,H,W])
  maps_rank<r>.npz    the fp32 form (ori [V,H,W,2], conf [V,H,W])
  <out>/files/{best_ori,conf,Ori}/<view>.png   batch_generate's files (written by the owning ranks)

    python tests/gabor_ranks_helper.py --out DIR --views 7
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def images(V, H=150, W=100):
    """V distinct gray uint8 views: stripes whose direction changes with the view, noise, a flat border"""
    out = []
    r, c = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for v in range(V):
        rng = np.random.default_rng(100 + v)
        th = np.pi * v / V
        im = 127 + 70 * np.cos(2 * np.pi * (r * np.cos(th) + c * np.sin(th)) / 4.0) + rng.normal(0, 6, (H, W))
        im[:10] = 30
        out.append(im.clip(0, 255).astype(np.uint8))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--views", type=int, default=7)
    a = ap.parse_args()
    import torch

    rank = 0
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as tdist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("MH_DIST_BACKEND", "nccl") == "nccl":
            tdist.init_process_group(backend="nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
        else:
            tdist.init_process_group(backend=os.environ["MH_DIST_BACKEND"])
        rank = tdist.get_rank()
    local = int(os.environ.get("MH_DEVICE_OVERRIDE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    from PIL import Image

    from monohair_amd import dist as mdist
    from monohair_amd.gabor import batch_generate, orientation_maps_device

    ims = images(a.views)
    k8, c8 = orientation_maps_device(ims, return_codes=True)
    ori, conf = orientation_maps_device(ims)
    os.makedirs(a.out, exist_ok=True)
    np.savez(os.path.join(a.out, "codes_rank%d.npz" % rank), k8=k8.cpu().numpy(), c8=c8.cpu().numpy())
    np.savez(os.path.join(a.out, "maps_rank%d.npz" % rank), ori=ori.cpu().numpy(), conf=conf.cpu().numpy())
    root = os.path.join(a.out, "files")
    if rank == 0:
        os.makedirs(os.path.join(root, "capture_images"), exist_ok=True)
        for v, im in enumerate(ims):
            Image.fromarray(im).save(os.path.join(root, "capture_images", "%03d.png" % v))
    mdist.barrier()
    batch_generate(root, "capture_images")
    mdist.barrier()


if __name__ == "__main__":
    main()
