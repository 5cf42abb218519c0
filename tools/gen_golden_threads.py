"""Golden vectors of the reference's forward() at OTHER thread counts (run here, from /root/reference, CPU torch):

    python tools/gen_golden_threads.py [threads ...]        -> tests/golden/pmvo_threads.npz   (default: 1 2 4)

Every other fixture was generated with torch's default 8 threads of the build container.  One rounding rule of the
reference depends on the thread count of the host it runs on: the column count from which MKL's sgemm evaluates
Camera.reprojection's [3,3] x [3, 90*M] product (/root/reference/Utils/Camera_utils.py:81-106) as an fma chain is the
switch to its THREADED kernel (8 threads: 28 445 columns = groups of >= 317 points; 4: 14 223; 2: 21 334; 1: never --
tools/probe_mkl_forms.py).  The kernels take it as the option `reproject_fma_min_cols`; this fixture pins that the option
reproduces the reference END TO END at those thread counts: forward() (/root/reference/PMVO.py:39-78) on one full
5000-point batch -- the first chunk of tests/golden/e2e_multichunk.npz, where 17-26 % of the (rank, point) items sit in
groups of more than 316 points and the largest group holds > 2000 -- under torch.set_num_threads(T), with the switch
column count probed at T in the same process stored next to the outputs.  The tests set the option from the fixture's
meta (oracle on the CPU, the HIP path under -m gpu) and assert every row.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)

from ref_import import import_reference  # noqa: E402
from monohair_amd import synth  # noqa: E402
import probe_mkl_forms as probe  # noqa: E402
from gen_golden_multichunk import CASE  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ROWS = 5000


def main():
    threads = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    R = import_reference()
    ref = R["PMVO"]
    case = CASE
    scene = synth.make_scene(case["V"], case["H"], case["W"], seed=case["seed"], scale=case["scale"], rings=case["rings"],
                             quantize=case["quantize"])
    cams = {}
    for c in scene["cams"]:
        cams[c["file"]] = R["Camera_utils"].Camera(c["ndc_prj"], np.linalg.inv(np.array(c["pose"])), c["file"])
    depths, Ori, Conf, masks = synth.scene_to_reference_dicts(scene)
    pm = ref.PMVO(cams, depths, Ori, Conf, masks, device="cpu", image_size=[case["H"], case["W"]],
                  patch_size=case["patch"], visible_threshold=case["vis_thr"], conf_threshold=case["thr"])
    ref.device = "cpu"
    mc = np.load(os.path.join(OUT, "e2e_multichunk.npz"))
    assert str(mc["meta"]) == repr(case)
    pts = mc["opt_select_p"][:ROWS].astype(np.float64)        # what optimize hands forward() for its first chunk
    out, meta = {}, dict(case=case, rows=ROWS, source="e2e_multichunk.npz: opt_select_p[:5000]", torch=torch.__version__,
                         default_threads=torch.get_num_threads(), by_threads={})
    g = torch.Generator().manual_seed(1)
    nt0 = torch.get_num_threads()
    for T in threads:
        cols = probe.chain_switch_columns(g, T)
        torch.set_num_threads(T)
        t0 = time.time()
        _, so, sl, sh = pm.forward(pts.copy())
        dt = time.time() - t0
        torch.set_num_threads(nt0)
        out["t%d_ori" % T], out["t%d_loss" % T], out["t%d_hc" % T] = so.numpy(), sl.numpy(), sh.numpy()
        meta["by_threads"][T] = dict(reproject_fma_min_cols=cols if cols else probe.NEVER, seconds=round(dt, 1))
        same = all(np.array_equal(out["t%d_%s" % (T, k)], mc["opt_" + n][:ROWS], equal_nan=True)
                   for k, n in (("ori", "select_o"), ("loss", "min_loss"), ("hc", "high_conf_index")))
        rows = int((~((out["t%d_ori" % T] == mc["opt_select_o"][:ROWS]) | np.isnan(out["t%d_ori" % T])).all(1)).sum())
        print("threads %d: switch at %s columns, forward %.1f s; equal to the 8-thread files: %s (%d rows differ)"
              % (T, cols or "never", dt, same, rows), flush=True)
        meta["by_threads"][T]["rows_differing_from_8_threads"] = rows
    np.savez_compressed(os.path.join(OUT, "pmvo_threads.npz"), meta=np.array(repr(meta)), **out)
    print("pmvo_threads.npz written:", meta["by_threads"])


if __name__ == "__main__":
    main()
