import sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
from monohair_amd import synth, _lib
from monohair_amd.camera import camera_records, cameras_from_list
from monohair_amd.pmvo import PMVO
dev = torch.device("cuda", 0)
scene = synth.make_scene(60, 1920, 1080, device=dev, seed=0, quantize=False)
cams = cameras_from_list(scene["cams"])
pm = PMVO.from_planes(camera_records(cams), scene["depth"], scene["ori"], scene["conf"], scene["mask"], device=dev,
                      patch_size=7, visible_threshold=1, conf_threshold=0.15, camera=cams)
cand = synth.candidate_points(res=256, seed=0)
cd = torch.from_numpy(np.ascontiguousarray(cand, dtype=np.float32)).to(dev)
M = cd.shape[0]
L = _lib.lib(); st = _lib.stream_ptr()
outs = [torch.empty(M, dtype=torch.uint8, device=dev) for _ in range(4)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {}
for rows in (1, 0, 1, 0):
    pm.set_option("filter_rows", rows)
    for which, ptrs in (("surface+filter", (outs[0], outs[1], None, None)), ("head only", (None, None, None, outs[3]))):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize(); ev[0].record()
            _lib.check(L.mh_filter_points(pm._ctx, _lib.ptr(cd), M, pm._side, 0.15, 1.0, *[_lib.ptr(p) for p in ptrs], M // 30, 0, M, st))
            ev[1].record(); torch.cuda.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]))
        print("filter_rows=%d %-15s %.3f ms" % (rows, which, best))
        res[(rows, which)] = [o.clone() for o in outs]
for which in ("surface+filter", "head only"):
    a, b = res[(1, which)], res[(0, which)]
    print(which, "equal:", all(torch.equal(x, y) for x, y in zip(a, b)))

# ---- experiment: the same launch on the points in BRICK order (4x4x4-voxel cubes of the candidate grid): does spatial coherence
# of a wave's 64 points cut the time?  (results permuted; only the time matters here)
lo = cd.min(0).values
vox = 0.005 / 2
for shift in (1, 2, 3):
    q = ((cd - lo) / vox).floor().long().clamp(0, 1023)
    b = q >> shift
    key = (b[:, 0] * 1024 + b[:, 1]) * 1024 + b[:, 2]
    perm = torch.argsort(key, stable=True)
    cp = cd[perm].contiguous()
    for rows in (1, 0):
        pm.set_option("filter_rows", rows)
        for which, ptrs in (("surface+filter", (outs[0], outs[1], None, None)), ("head only", (None, None, None, outs[3]))):
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); ev[0].record()
                _lib.check(L.mh_filter_points(pm._ctx, _lib.ptr(cp), M, pm._side, 0.15, 1.0, *[_lib.ptr(p) for p in ptrs], M // 30, 0, M, st))
                ev[1].record(); torch.cuda.synchronize()
                best = min(best, ev[0].elapsed_time(ev[1]))
            print("bricks of %d^3 voxels, filter_rows=%d %-15s %.3f ms" % (1 << shift, rows, which, best))
