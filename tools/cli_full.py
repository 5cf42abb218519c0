#!/usr/bin/env python
"""End-to-end wall time of the drop-in command line at the headline size: writes a synthetic capture (60 views @
1920x1080 in the reference's on-disk layout) and runs `python PMVO.py --yaml=...` on it in fresh processes, first from
the file tree, then from a maps pack (written by the first pack run, read by the second), with MH_TIMING=1.
    python tools/cli_full.py [--views 60] [--size 1920 1080] [--res 128]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monohair_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=60)
ap.add_argument("--size", type=int, nargs=2, default=[1920, 1080])
ap.add_argument("--res", type=int, default=128, help="tessellation of the COLMAP stand-in sphere (sets the candidate count)")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="mhcli_")
t0 = time.time()
synth.write_case(tmp, "synthetic_sphere", V=a.views, H=a.size[0], W=a.size[1], res=a.res)
print("wrote the capture in %.1f s" % (time.time() - t0), flush=True)
common = [sys.executable, os.path.join(ROOT, "PMVO.py"), "--yaml=configs/reconstruct/synthetic_sphere",
          "--data.root=%s" % tmp, "--data.image_size=[%d,%d]" % tuple(a.size), "--PMVO.patch_size=7"]
env = dict(os.environ, PYTHONPATH=ROOT, MH_TIMING="1")
for name, extra in (("tree", []), ("pack(write)", ["--data.maps_pack=maps.mhpk"]), ("pack(read)", ["--data.maps_pack=maps.mhpk"])):
    t0 = time.time()
    r = subprocess.run(common + ["--name=%s" % name.split("(")[0]] + extra, cwd=ROOT, env=env, stdin=subprocess.DEVNULL,
                       capture_output=True, text=True)
    dt = time.time() - t0
    print("== %s: exit %d, %.2f s wall (python start-up + torch import included)" % (name, r.returncode, dt))
    for line in (r.stdout + r.stderr).splitlines():
        if "[mh-timing]" in line or "points:" in line or "surface_num" in line or "Error" in line:
            print("   ", line.strip())
