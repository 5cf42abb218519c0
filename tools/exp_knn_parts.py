#!/usr/bin/env python
"""Where do the 7 ms of refine's surface k-NN go?  (GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monohair_amd import synth
from monohair_amd.pmvo_utils import GridKNN

cand = synth.candidate_points(res=256, seed=0)
rng = np.random.default_rng(0)
r = np.linalg.norm(cand, axis=1)
pts = cand[np.argsort(np.abs(r - np.median(r)))[:287696]].astype(np.float32)
dev = "cuda:0"
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = GridKNN(pts, k_hint=100, device=dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    idx = g.query(pts, 100, int32=True, self_query=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("init %.2f ms, query %.2f ms (retries %d), h=%.5f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, g.last_retries, g.h))
