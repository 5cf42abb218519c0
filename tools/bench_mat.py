#!/usr/bin/env python
"""Time the sparse MAT-v5 writer (monohair_amd.pmvo_utils.save_ori_occ_mat_sparse -> mh_mat_write_sparse) on a sphere
shell of voxels at the reference's 256 x 256 x 192 grid, for several thread counts and target directories."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import pmvo_utils as U

g = np.mgrid[0:256, 0:256, 0:192].reshape(3, -1).T
c = g * U.VOXEL_SIZE + U.VOXEL_MIN
v = g[np.abs(np.linalg.norm(c, axis=1) - 0.12) < 0.0035]
o = np.random.default_rng(0).normal(size=(len(v), 3))
print("voxels", len(v))
for base in ("/tmp", "/dev/shm", os.getcwd()):
    if not os.path.isdir(base):
        continue
    for th in (1, 4, 16, 32):
        ts = []
        for rep in range(3):
            d = tempfile.mkdtemp(dir=base)
            t = time.perf_counter()
            U.save_ori_occ_mat_sparse(d, U.GRID_RESOLUTION, v, o, threads=th)
            ts.append(time.perf_counter() - t)
            for f in os.listdir(d):
                os.remove(os.path.join(d, f))
            os.rmdir(d)
        print("%-10s threads %2d  %.1f ms (min of 3)" % (base, th, min(ts) * 1e3))
