#!/usr/bin/env python
"""Experiment (GPU box): where do the sporadic slow rounds of the 8-bit loop come from?
    python tools/exp_host_numa.py any|<numa node>        host-only us per forward() (tiny chunks) and 8-bit iterations/s
    EXP_ROUNDS=40 [EXP_GC=0] python tools/exp_host_numa.py any|<node>   per-round rates; for slow rounds the slowest calls
Findings (round 3): the steady rate is the same bound to one NUMA node or not, with or without the Python GC (4 190 it/s);
a pinned-slot ring that grew slot by slot caused slow rounds while it grew (now one slab per stream); what remains is ~1 round
in 40 in which ONE call blocks ~6 ms with no allocation anywhere: the host (30 iterations ahead) waiting on a device pause."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def node_of_cpu(c):
    for d in glob.glob("/sys/devices/system/node/node*"):
        lst = open(d + "/cpulist").read().strip()
        for part in lst.split(","):
            lo, _, hi = part.partition("-")
            if int(lo) <= c <= int(hi or lo):
                return int(d.rsplit("node", 1)[1])
    return -1


def gpu_nodes():
    out = []
    for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
        try:
            out.append(int(open(f).read()))
        except Exception:
            pass
    return out


if len(sys.argv) > 1 and sys.argv[1] != "any":
    node = int(sys.argv[1])
    lst = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    cpus = set()
    for part in lst.split(","):
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    os.sched_setaffinity(0, cpus)
from monohair_amd import synth
from monohair_amd.camera import cameras_from_list
from monohair_amd.pmvo import PMVO

dev = "cuda:0"
V, H, W = 60, 1920, 1080
sc = synth.make_scene_codes(V, H, W, device=dev, seed=0)
pm = PMVO.from_u8(cameras_from_list(sc["cams"]), sc["depth"], sc["ori_u8"], sc["conf_u8"], sc["mask_u8"], device=dev,
                  image_size=[H, W], patch_size=7, visible_threshold=1, conf_threshold=0.15)
cand = synth.candidate_points(res=256, seed=0)
chunks = [cand[i * 5000:(i + 1) * 5000].astype(np.float32) for i in range(24)]
small = [c[:64] for c in chunks]
streams = pm.side_streams(3)


def loop(cs, n):
    for i in range(n):
        with torch.cuda.stream(streams[i % 3]):
            pm.forward(cs[i % len(cs)])


loop(chunks, 60)
torch.cuda.synchronize()
if os.environ.get("EXP_GC") == "0":
    import gc

    gc.collect()
    gc.freeze()
    gc.disable()
if os.environ.get("EXP_ROUNDS"):
    r = []
    for rd in range(int(os.environ["EXP_ROUNDS"])):
        t0 = time.perf_counter()
        calls = []
        for i in range(100):
            ta = time.perf_counter()
            with torch.cuda.stream(streams[i % 3]):
                pm.forward(chunks[i % len(chunks)])
            calls.append(time.perf_counter() - ta)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r.append(int(100 / (t2 - t0)))
        ms = torch.cuda.memory_stats()
        nda = (ms.get("num_device_alloc", -1), ms.get("num_device_free", -1), ms.get("segment.all.current", -1))
        try:
            hs = torch.cuda.host_memory_stats()
            nha = (hs.get("num_host_alloc", -1), hs.get("num_host_free", -1))
        except Exception:
            nha = None
        if r[-1] < 3900:
            print("  device allocs/frees/segments %s host %s" % (nda, nha))
            top = sorted(calls)[-3:]
            print("  slow round %d: %d it/s; enqueue %.1f ms, drain %.1f ms; slowest calls (ms) %s; ring %s" % (
                rd, r[-1], (t1 - t0) * 1e3, (t2 - t1) * 1e3, [round(x * 1e3, 2) for x in top],
                [v["i"] for v in pm._stage.values()]))
    print({"allowed": len(os.sched_getaffinity(0)), "gc": os.environ.get("EXP_GC", "1"), "rounds": r, "end": (nda, nha)})
    sys.exit(0)
res = {}
for name, cs in (("host_us", small), ("it_s", chunks)):
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        loop(cs, 200)
        torch.cuda.synchronize()
        best.append(time.perf_counter() - t0)
    dt = sorted(best)[1]
    res[name] = round(dt / 200 * 1e6, 1) if name == "host_us" else round(200 / dt, 1)
cpu = os.sched_getcpu() if hasattr(os, "sched_getcpu") else -1
print({"cpu": cpu, "node": node_of_cpu(cpu), "gpu_nodes": gpu_nodes(), "ncpu_allowed": len(os.sched_getaffinity(0)), **res})
