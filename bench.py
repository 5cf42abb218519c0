#!/usr/bin/env python
"""bench.py -- PMVO iterations/sec on the synthetic 60-view 1080p / 256^3 workload (BASELINE.json).

One step = one PMVO iteration = one `PMVO.forward()` over a chunk of 5000 candidate points against all
V views (one step of the trange at /root/reference/PMVO.py:572-574): project-and-gather, base-view
ranking, tap preparation and the fused loss search, maps resident in HBM, no file IO.

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

Points shard across ranks (every GPU holds all views; no collective inside an iteration -- SURVEY.md §8e),
so per-GPU work is fixed as N grows: "scaling": "weak".  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from monohair_amd import synth  # noqa: E402
from monohair_amd.camera import camera_records, cameras_from_list  # noqa: E402
from monohair_amd.pmvo import PMVO, depth_offsets  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CHUNK = 5000                 # PMVO.py:566


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views", type=int, default=60)
    ap.add_argument("--height", type=int, default=1920)
    ap.add_argument("--width", type=int, default=1080)
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--patch", type=int, default=7)          # big_wavy1.yaml:18
    ap.add_argument("--conf-threshold", type=float, default=0.15)
    ap.add_argument("--quantize", action="store_true", help="8-bit orientation/confidence maps (file hand-off)")
    ap.add_argument("--cpu-points", type=int, default=0, help="points of the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra 8-bit-maps measurement")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the independent iterations alternate on (as monohair_amd.pmvo.optimize does)")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: MH_DEVICE_OVERRIDE pins every rank to one GPU and MH_DIST_BACKEND=gloo replaces RCCL, so that the
    # world_size > 1 code path can be exercised on a single-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("MH_DEVICE_OVERRIDE") is not None:
        local = int(os.environ["MH_DEVICE_OVERRIDE"])
    backend = os.environ.get("MH_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    n_gpus = max(a.gpus, world) if world > 1 else 1
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    V, H, W, P = a.views, a.height, a.width, a.patch * a.patch
    scene = synth.make_scene(V, H, W, device=dev, seed=0, quantize=a.quantize)
    cams = cameras_from_list(scene["cams"])
    recs = camera_records(cams)
    pm = PMVO.from_planes(recs, scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                          patch_size=a.patch, visible_threshold=1, conf_threshold=a.conf_threshold, camera=cams)
    if a.variant:
        pm.set_option("search_variant", a.variant)

    # candidate points of the 256^3 volume; keep the ones the reference would send to optimize()
    # (filter_negative_points, PMVO.py:535-557), then chunk by 5000 and deal the chunks to the ranks
    cand = synth.candidate_points(res=a.volume, seed=0)
    surf = []
    for i in range(0, len(cand), 200000):
        s, _, _ = pm.filter_points(cand[i:i + 200000])
        surf.append(s.cpu().numpy())
    surf = np.concatenate(surf)
    pts = cand[surf]
    nchunk = max(1, len(pts) // CHUNK)
    chunks = [pts[i * CHUNK:(i + 1) * CHUNK] for i in range(nchunk)]
    my = [chunks[i] for i in range(rank, nchunk, world)] or chunks[:1]
    dev_chunks = [torch.from_numpy(c).to(dev).float() for c in my]

    # consecutive iterations are independent chunks of `optimize` (PMVO.py:572-574); like the driver in
    # monohair_amd/pmvo.py they alternate between HIP streams so one chunk's tail overlaps the next one's head
    streams = pm.side_streams(max(1, a.streams))
    # set-up, not steps: the per-stream tap-list scratch (1.2 GB each) and the allocator pools of the per-iteration
    # outputs exist before anything is timed, whatever --warmup is
    for st in streams:
        with torch.cuda.stream(st):
            pm._get_scratch(CHUNK)
            hold = [torch.empty((V, CHUNK, 2), device=dev), torch.empty((V, CHUNK), device=dev),
                    torch.empty((V, CHUNK), device=dev), torch.empty((V, CHUNK), device=dev),
                    torch.empty((20, CHUNK), device=dev), torch.empty((20, CHUNK), device=dev, dtype=torch.int32),
                    torch.empty((CHUNK, 3), device=dev), torch.empty((CHUNK,), device=dev),
                    torch.empty((CHUNK,), device=dev, dtype=torch.bool)]
            del hold
            pm.forward(dev_chunks[0][:1])       # one point: loads the kernels' code objects (not a workload step)
    torch.cuda.synchronize()

    def step(i):
        with torch.cuda.stream(streams[i % len(streams)]):
            return pm.forward(dev_chunks[i % len(dev_chunks)])

    for i in range(a.warmup):
        step(i)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    # --- timed region: exactly K steps
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # --- per-kernel durations on the same chunks, outside the timed region: HIP events on the launch stream,
    # REP back-to-back launches per measurement so that launch gaps do not count as kernel time
    def timed(fn, rep):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(rep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / rep

    t_pg = t_tk = t_sr = nvis = 0.0
    reps = max(2, min(len(dev_chunks), 6))
    for i in range(reps):
        c = dev_chunks[i % len(dev_chunks)]
        t_pg += timed(lambda: pm.Compute_Visible_and_Ori(c), 10)
        t_tk += timed(lambda: pm.Find_max_conf_from_visible_view(), 10)
        nvis += float((pm.visible != -1).float().mean().item())
        t_sr += timed(lambda: pm.forward(c), 5)
    t_pg, t_tk, t_sr, nvis = t_pg / reps, t_tk / reps, t_sr / reps, nvis / reps
    t_search = max(t_sr - t_tk, 1e-6)     # forward = fused front end + top-k + search

    if rank != 0:
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
            dist.destroy_process_group()
        return
    ms = dt / a.steps * 1e3
    value = a.steps * world / dt
    # algorithmic bytes of project-and-gather per iteration: V*N*(12P+20) gathered + the same
    # written + 12N of points, fp32 reference layout
    pg_bytes = 2 * V * CHUNK * (12 * P + 20) + 12 * CHUNK
    achieved = pg_bytes / (t_pg * 1e-3) / 1e9
    pairs_nominal = 10 * V * CHUNK * 90 * P
    out = {
        "metric": "PMVO iterations/sec (60x1080p views, 256^3 volume)",
        "value": round(value, 3),
        "unit": "iterations/s",
        "n_gpus": n_gpus,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "synthetic sphere, %d views @ %dx%d, %d^3 volume, %d points/iteration, patch %d, S=90, "
                        "10 base views (%s)" % (V, H, W, a.volume, CHUNK, a.patch, {
                            (60, 1920, 1080, 256): "BASELINE.json configs[2], the configuration the metric is quoted on",
                            (30, 1920, 1080, 128): "BASELINE.json configs[1]",
                            (120, 3840, 2160, 512): "BASELINE.json configs[4], one GPU's share"}.get(
                                (V, H, W, a.volume), "custom size")),
            "views": V, "image": [H, W], "volume": a.volume, "points_per_iteration": CHUNK, "patch": a.patch,
            "conf_threshold": a.conf_threshold, "surface_points": int(len(pts)), "iterations_full_pass": nchunk,
            "parallelism": "points sharded over %d GPU(s), views replicated" % world,
            "maps": "quantized-8bit" if a.quantize else "continuous",
            "streams": len(streams),
        },
        "roofline": {
            "kernel": "mh_project_gather_kernel<%d>" % a.patch,
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": pmc_traffic("mh_project_gather_kernel<%d>" % a.patch, V, H, W),
            "algorithmic_bytes_per_launch": pg_bytes,
            "launch_ms": round(t_pg, 4),
        },
        "kernels_ms": {"forward_total_single_stream": round(t_sr, 4), "topk": round(t_tk, 4),
                       "project_taps+search": round(t_search, 4),
                       "project_gather_api_kernel": round(t_pg, 4)},
        "search": {
            "pair_evals_nominal": pairs_nominal,
            "visible_view_fraction": round(nvis, 4),
            "gpair_per_s_nominal": round(pairs_nominal / (t_search * 1e-3) / 1e9, 1),
        },
    }
    if not a.quantize and not a.no_secondary and world == 1:
        try:
            out["secondary_8bit_maps"] = secondary_quantized(a, dev, recs, cams, dev_chunks)
        except Exception as e:   # the secondary number must never cost the headline line
            out["secondary_8bit_maps"] = {"error": repr(e)[:200]}
    if not a.no_secondary and world == 1:
        try:
            out["secondary_full_pass"] = secondary_full_pass(dev, pm, cand)
        except Exception as e:
            out["secondary_full_pass"] = {"error": repr(e)[:200]}
        try:
            out["secondary_gabor_bank"] = secondary_gabor(a, dev)
        except Exception as e:
            out["secondary_gabor_bank"] = {"error": repr(e)[:200]}
    # last: its 128 OpenMP workers keep spinning for a while after the parallel region and would slow the host side
    # of the secondary legs
    if not a.no_cpu and world == 1:      # the CPU leg runs on rank 0 at N=1 only
        try:
            out["cpu_baseline"] = cpu_baseline(a, scene, recs, my[0], ms)
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)[:200]}
    print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def secondary_quantized(a, dev, recs, cams, dev_chunks):
    """The same iteration on maps pushed through the reference's 8-bit file hand-off (integer degrees, conf/255 --
    what a real capture delivers): duplicate tap orientations are dropped exactly."""
    scene_q = synth.make_scene(a.views, a.height, a.width, device=dev, seed=0, quantize=True)
    pm = PMVO.from_planes(recs, scene_q["depth"], scene_q["ori"], scene_q["conf"], scene_q["mask"], device=dev,
                          patch_size=a.patch, visible_threshold=1, conf_threshold=a.conf_threshold, camera=cams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, a.streams))]

    def step(i):
        with torch.cuda.stream(streams[i % len(streams)]):
            return pm.forward(dev_chunks[i % len(dev_chunks)])

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(a.steps / dt, 3), "unit": "iterations/s", "ms_per_step": round(dt / a.steps * 1e3, 4),
            "maps": "quantized-8bit"}


def secondary_full_pass(dev, pm, cand):
    """Wall time of the whole exterior pass on this scene: filter_negative_points -> optimize ->
    refine (smoothing, shell points, voxel fit, Ori3D.mat / Occ3D.mat written), one process, stages synchronised."""
    import tempfile
    import types

    from scipy.spatial import KDTree

    from monohair_amd.pmvo import filter_negative_points, optimize, refine

    rng = np.random.default_rng(1)
    b = rng.normal(size=(2000, 3))
    b = b / np.linalg.norm(b, axis=1, keepdims=True) * 0.09          # stand-ins for the bust / scalp meshes
    scalp = b[b[:, 1] > 0.03] * (0.1 / 0.09)
    pm.set_head(KDTree(b), KDTree(scalp), scalp.max(0))
    tmp = tempfile.mkdtemp(prefix="mhbench_")
    args = types.SimpleNamespace(device=str(dev), output_path=tmp, save_root=tmp + "/optimize", save_path=tmp + "/refine",
                                 PMVO=types.SimpleNamespace(visible_threshold=1), data=types.SimpleNamespace(root=tmp))
    os.makedirs(args.save_path, exist_ok=True)
    T = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        T[name] = round(time.perf_counter() - t0, 3)
        return r

    import contextlib
    import io

    with contextlib.redirect_stdout(io.StringIO()):                   # the drivers print progress like the reference
        s_idx, s_pts, f_idx = timed("filter_s", lambda: filter_negative_points(cand, pm, args))
        sp, so, ml, _ = timed("optimize_s", lambda: optimize(s_pts, pm, args))
        timed("refine_and_volume_s", lambda: refine(sp, so, ml, pm, cand[:len(f_idx)][f_idx].astype(np.float32), args,
                                                   infer_inner=False, threshold=0.025, return_dense=False))
    T["total_s"] = round(sum(T.values()), 3)
    T.update(candidates=int(len(cand)), surface_points=int(s_idx.sum()), shell_points=int(f_idx.sum()),
             iterations=int(len(s_pts) // CHUNK + 1), unit="s")
    return T


def secondary_gabor(a, dev):
    """The other kernel family of the path: the per-view Gabor orientation bank on one
    synthetic view of the benchmark's image size, FP32-MFMA im2col kernel, HIP events around 10 launches.
    Compute bound: 2*180*289 FLOP per pixel against 12 B; peak = 157.3 TFLOP/s dense fp32 matrix."""
    import numpy as np

    from monohair_amd.gabor import calOrientationGabor

    H, W = a.height, a.width
    g = torch.Generator(device="cpu").manual_seed(0)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    img = (0.25 * torch.cos(2 * np.pi * (0.6 * xx + 0.8 * yy) / 4.0) + 0.02 * torch.randn((H, W), generator=g)).float().to(dev)
    gab = calOrientationGabor(device=dev)
    gab.filter_index(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gab.filter_index(img)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tf = 2.0 * 180 * 289 * H * W / ms / 1e9
    return {"metric": "Gabor bank views/s", "value": round(1e3 / ms, 1), "unit": "views/s", "ms_per_view": round(ms, 3),
            "image": [H, W], "kernel": "mh_gabor_mfma_kernel",
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": 157.3, "unit": "TFLOP/s",
                         "frac": round(tf / 157.3, 4), "traffic": None}}


def pmc_traffic(kernel, V, H, W):
    """HBM bytes per launch of the roofline kernel from the committed rocprofv3 PMC pass (profiles/traffic.json;
    counters cannot be read from inside this process).  None when no pass exists for this workload."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if t.get("workload", "").startswith("%d views @ %dx%d" % (V, H, W)):
            return t[kernel]["traffic_bytes"]
    except Exception:
        pass
    return None


def cpu_baseline(a, scene, recs, chunk, gpu_ms):
    """The CPU oracle (oracle/pmvo_oracle.c, OpenMP over points) timed on this host on a bounded sample of the
    same iteration: the first n points of the chunk against all views."""
    import oracle

    views = oracle.Views(recs, scene["depth"].cpu().numpy(), scene["ori"].cpu().numpy(),
                         scene["conf"].cpu().numpy(), scene["mask"].cpu().numpy())
    cores = oracle.num_threads()
    offs = depth_offsets(90)
    n = a.cpu_points if a.cpu_points > 0 else len(chunk)
    # repeat the sample until >= 12 s of CPU work have been timed (bounded: at most 40 repeats)
    t, reps = 0.0, 0
    while reps < 40 and (t < 12.0 or reps == 0):
        t0 = time.perf_counter()
        oracle.forward(views, chunk[:n], a.patch, a.conf_threshold, offs)
        t += time.perf_counter() - t0
        reps += 1
    n_total = n * reps
    its = (n_total / float(CHUNK)) / t
    return {
        "value": round(its, 5),
        "unit": "iterations/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d x %d points of one %d-point iteration, all %d views, C oracle (oracle/pmvo_oracle.c) with "
                  "OpenMP on %d threads, %.1f s of wall time" % (reps, n, CHUNK, a.views, cores, t),
    }


if __name__ == "__main__":
    main()
