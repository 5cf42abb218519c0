"""How often does the key body of mh_search3_kernel evaluate a view twice?  (round 4)

Builds a copy of the library with -DMH_KEY_STATS (counters in mh_search_slices_lds), runs bench.py's timed loop against it
and prints (wave, view) visits: all / one-tap or NaN-seed lists (select body directly) / re-evaluated after the key body.

    python tools/exp_key_stats.py [bench.py arguments, e.g. --quantize]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "monohair_amd", "csrc")
LIB = os.path.join(ROOT, "monohair_amd", "lib", "libmhpmvo_keystats.so")


def build():
    objs = [o for o in sorted(os.listdir(CSRC)) if o.endswith(".o") and o != "pmvo_search.o"]
    so = os.path.join(CSRC, "pmvo_search_keystats.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-DMH_KEY_STATS", "-c", os.path.join(CSRC, "pmvo_search.hip"), "-o", so])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, so] +
                          [os.path.join(CSRC, o) for o in objs if o != "pmvo_search_keystats.o"] + ["-ldl"])


if __name__ == "__main__":
    if "--build-only" in sys.argv:
        build()
        sys.exit(0)
    build()   # (always: a stale copy would count another kernel)
    sys.path.insert(0, ROOT)
    from monohair_amd import _lib

    _lib.LIB_PATH = LIB
    sys.argv = ["bench.py", "--no-cpu", "--no-secondary", "--steps", "57", "--warmup", "0"] + sys.argv[1:]
    import bench

    bench.main()
    h = ctypes.CDLL(LIB)
    out = (ctypes.c_ulonglong * 4)()
    assert h.mh_debug_key_stats(out, 0) == 0
    tot, direct, again = out[0], out[1], out[2]
    print({"wave_view_visits": tot, "select_body_directly": direct, "re_evaluated": again,
           "re_evaluated_fraction": again / max(1, tot)})
