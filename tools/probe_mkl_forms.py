"""How the reference's CPU matmuls and sums round, by size -- probed on THIS machine, and recorded (round 5).

PMVO.sample_next_3d_pos (/root/reference/PMVO.py:263-335) calls, per camera, Camera.projection and Camera.reprojection
(/root/reference/Utils/Camera_utils.py:38-58, 81-106) on the M points of the batch whose base view that camera is:
    torch.matmul(pose [4,4], vertices [4,M]),  torch.matmul(proj [4,4], camera_v [4,M])           (:50,53)
    torch.matmul(torch.linalg.inv(pose[:3,:3]) [3,3], (camera_v[:3] - t) [3, S*M])                 (:103)
and PMVO.compute_prj_loss sums [V, N, S] tensors over dim 0 (PMVO.py:198-204).  Which MKL sgemm kernel a product lands in --
and so the association of its fp32 sums -- depends on the number of columns (and, for the switch to MKL's threaded kernel,
on the number of threads); ATen's outer sum adds the trailing columns of a tensor in another order than the rest.  The
oracle (oracle/pmvo_oracle.c: mm4_elem, mm3_elem, row_sum1) and the kernels (csrc/mh_device.h: MhRule) restate what this
script finds.  It

  1. classifies every column of those products against every expression tree over the K products (separately rounded
     or fused), for M = 1 .. and around the switch, and prints the table;
  2. bisects the column count at which the [3,3] x [3,C] product switches to the chain form, for 1..8 threads;
  3. classifies the columns of torch.sum(dim=0) (cascade / row_sum);
  4. writes tests/golden/mkl_forms.npz: inputs and torch's outputs at the boundary sizes, with torch's version, thread
     count and build string in `meta` -- tests/test_oracle_forms.py pins the oracle's forms to these recorded outputs.

The thresholds are facts about MKL 2024.2 / AVX-512 / 8 threads (where every golden under tests/golden/ was generated);
another host may switch elsewhere: options reproject_fma_min_cols (kernels) / oracle.set_reproject_rule(...).

    python tools/probe_mkl_forms.py [--quick]      (build container; CPU torch)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LD = np.longdouble


def fma(a, b, c):
    # exact product in 64-bit-mantissa arithmetic, one rounding of the sum to that, then to float32 (probe only)
    return (np.asarray(a).astype(LD) * np.asarray(b).astype(LD) + np.asarray(c).astype(LD)).astype(np.float32)


def tree_forms(a, B):
    """every way to evaluate sum_k a[k] * B[k] in fp32: products rounded separately or fused into an addition"""
    K = len(a)
    full = (1 << K) - 1
    V = {1 << k: {"p%d" % k: (a[k] * B[k]).astype(np.float32)} for k in range(K)}
    for S in range(1, full + 1):
        if S in V:
            continue
        out = {}
        sub = (S - 1) & S
        while sub:
            oth = S ^ sub
            if sub < oth:
                for n1, v1 in V[sub].items():
                    for n2, v2 in V[oth].items():
                        out["(%s+%s)" % (n1, n2)] = (v1 + v2).astype(np.float32)
            sub = (sub - 1) & S
        for k in range(K):
            if S >> k & 1:
                for n1, v1 in V[S ^ (1 << k)].items():
                    out["f%d[%s]" % (k, n1)] = fma(np.float32(a[k]), B[k], v1)
        V[S] = out
    return V[full]


def classify(A, B, O):
    """names of the forms that reproduce EVERY element of O = A @ B"""
    ok = None
    for r in range(A.shape[0]):
        f = tree_forms(A[r], B)
        s = {n for n, v in f.items() if np.array_equal(v, O[r], equal_nan=True)}
        ok = s if ok is None else ok & s
    return ok


def mm3_inputs(g, C):
    """operands with the strides the reference hands to torch.matmul at Camera_utils.py:103"""
    R = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    t = torch.randn(3, 1, generator=g)
    A = torch.linalg.inv(R)                                   # LAPACK output: column-major
    B = torch.randn(C, 4, generator=g).permute(1, 0)[:3] - t  # a transposed view minus a column
    return A, B


def mm4_inputs(g, M):
    A = torch.randn(4, 4, generator=g)
    v = torch.randn(M, 3, generator=g).permute(1, 0)
    B = torch.cat([v, torch.ones((1, M))])                    # Camera_utils.py:48-49
    return A, B


def multi_row_sum(x):
    """ATen's multi_row_sum over the rows of x [R, C] (SumKernel.cpp), vectorised over the columns"""
    R, C = x.shape
    clog = 0 if R <= 1 else int(R - 1).bit_length()
    lp = max(4, clog // 4)
    step, mask = 1 << lp, (1 << lp) - 1
    acc = [np.zeros(C, np.float32) for _ in range(4)]
    i = 0
    while i + step <= R:
        for _ in range(step):
            acc[0] = acc[0] + x[i]
            i += 1
        for j in range(1, 4):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = np.zeros(C, np.float32)
            if (i & (mask << (j * lp))) != 0:
                break
    while i < R:
        acc[0] = acc[0] + x[i]
        i += 1
    for j in range(1, 4):
        acc[0] = acc[0] + acc[j]
    return acc[0]


def row_sum(x):
    R = x.shape[0]
    L = R // 4
    parts = [multi_row_sum(x[k:L * 4:4]) for k in range(4)]
    for i in range(L * 4, R):
        parts[0] = parts[0] + x[i]
    for k in range(1, 4):
        parts[0] = parts[0] + parts[k]
    return parts[0]


def chain_switch_columns(g, nt, hi=400000):
    """The column count from which torch.matmul([3,3], [3,C]) takes the fma-chain form with `nt` threads (0: it never
    does up to `hi` columns) -- the value of the option reproject_fma_min_cols for a reference host with that many threads.
    Restores the thread count."""
    nt0 = torch.get_num_threads()
    torch.set_num_threads(nt)
    try:
        def chain(C):
            A, B = mm3_inputs(g, C)
            f = classify(A.numpy().copy(), B.numpy().copy(), torch.matmul(A, B).numpy())
            return "f2[f1[p0]]" in f and "(p1+(p0+p2))" not in f

        lo = 90
        if not chain(hi):
            return 0
        while hi - lo > 1:
            mid = (lo + hi) // 2
            lo, hi = (lo, mid) if chain(mid) else (mid, hi)
        return hi
    finally:
        torch.set_num_threads(nt0)


NEVER = 2 ** 31 - 1     # reproject_fma_min_cols for a host whose sgemm never switches (one thread)


def emit_options(threads=None):
    """`python tools/probe_mkl_forms.py --emit-options [--threads N]` on the host the REFERENCE runs on: prints the JSON the
    drivers take as --PMVO.reference_host=<file> (PMVO.py) so that the kernels round as that host's MKL does."""
    import json

    nt = torch.get_num_threads() if threads is None else int(threads)
    g = torch.Generator().manual_seed(1)
    cols = chain_switch_columns(g, nt)
    mkl = [l.strip() for l in torch.__config__.show().split("\n") if "Math Kernel" in l]
    print(json.dumps({"reproject_fma_min_cols": cols if cols else NEVER, "sum_block": 32, "threads": nt,
                      "torch": torch.__version__, "mkl": (mkl[0][:150] if mkl else "")}))


def main():
    if "--emit-options" in sys.argv:
        emit_options(sys.argv[sys.argv.index("--threads") + 1] if "--threads" in sys.argv else None)
        return
    quick = "--quick" in sys.argv
    g = torch.Generator().manual_seed(1)
    nt0 = torch.get_num_threads()
    print("torch %s, %d threads" % (torch.__version__, nt0))
    out = {}

    print("\n[4,4] x [4,M]  (Camera.projection)")
    for M in (1, 2, 3, 5, 90, 5000, 450000):
        forms = None
        for trial in range(40 if M == 1 else 3):
            A, B = mm4_inputs(g, M)
            O = torch.matmul(A, B)
            f = classify(A.numpy(), B.numpy().copy(), O.numpy())
            forms = f if forms is None else forms & f
            if trial == 0 and M in (1, 2, 5):
                out["mm4_M%d_A" % M], out["mm4_M%d_B" % M], out["mm4_M%d_out" % M] = A.numpy(), B.numpy().copy(), O.numpy()
        print("  M = %-7d %s" % (M, sorted(forms)[:4]))

    print("\n[3,3] x [3,C]  (Camera.reprojection), %d threads" % nt0)
    for C in (1, 2, 3, 4, 5, 90, 180, 28440, 28444, 28445, 28530, 90000):
        forms = None
        for trial in range(20 if C <= 5 else 2):
            A, B = mm3_inputs(g, C)
            O = torch.matmul(A, B)
            f = classify(A.numpy().copy(), B.numpy().copy(), O.numpy())
            forms = f if forms is None else forms & f
            if trial == 0 and C in (1, 3, 4, 90, 28444, 28445):
                out["mm3_C%d_A" % C], out["mm3_C%d_B" % C], out["mm3_C%d_out" % C] = \
                    A.numpy().copy(), B.numpy().copy(), O.numpy()
        print("  C = %-7d %s" % (C, sorted(forms)[:4]))

    print("\nswitch of the [3,3] x [3,C] product to the chain form, by thread count")
    switch = {}
    for nt in ([nt0] if quick else range(1, nt0 + 1)):
        switch[nt] = chain_switch_columns(g, nt)
        if switch[nt]:
            print("  %d thread(s): chain form from %d columns (= %d points of 90 samples)" % (nt, switch[nt], -(-switch[nt] // 90)))
        else:
            print("  %d thread(s): no switch up to 400000 columns" % nt)
    torch.set_num_threads(nt0)

    print("\ntorch.sum(x [V, C], dim=0): columns that match ONLY the cascade / ONLY row_sum, before / in the last C mod 32")
    for V, C in ((24, 21600), (24, 21616), (24, 3600), (60, 450000), (300, 4320), (20, 270), (24, 376), (300, 72)):
        x = torch.rand(V, C, generator=g)
        t = torch.sum(x, dim=0).numpy()
        a, b = multi_row_sum(x.numpy()), row_sum(x.numpy())
        ts = C - C % 32
        ca, ro = (a == t) & (b != t), (b == t) & (a != t)
        print("  V=%-4d C=%-7d C mod 32 = %-3d head: cascade-only %d, row_sum-only %d | tail: cascade-only %d, row_sum-only %d, "
              "neither %d" % (V, C, C % 32, ca[:ts].sum(), ro[:ts].sum(), ca[ts:].sum(), ro[ts:].sum(),
                              ((a != t) & (b != t)).sum()))
        if (V, C) in ((24, 376), (300, 72), (20, 270)):
            out["sum_V%d_C%d_x" % (V, C)], out["sum_V%d_C%d_out" % (V, C)] = x.numpy(), t
    # torch.sqrt / x ** 0.5 is not the IEEE root in this build (float and double, any size, contiguous or not: 0.7 % of random
    # values are one ulp off -- ATen's unary ops go through MKL's vector math library, vsSqrt in high-accuracy mode: below 1 ulp) -- the one operation of the Gabor confidence
    # (`variance ** (1 / 2)`, preprocess_capture_data/GaborFilter.py:77) that is not restated (closed source)
    xs = torch.rand(4096, generator=g) * 3
    rs = torch.sqrt(xs)
    ieee = np.sqrt(xs.numpy())
    print("\ntorch.sqrt on 4096 random floats: %d differ from the IEEE root (all by one ulp: %s); x ** 0.5 == torch.sqrt: %s" % (
        int((rs.numpy() != ieee).sum()), bool(np.all(np.abs(rs.numpy().view(np.int32) - ieee.view(np.int32)) <= 1)),
        bool(torch.equal(xs ** 0.5, rs))))
    out["sqrt_x"], out["sqrt_out"] = xs.numpy(), rs.numpy()
    meta = dict(torch=torch.__version__, threads=nt0, mkl=[l.strip() for l in torch.__config__.show().split("\n") if "Math Kernel" in l][0][:150],
                fma_min_cols=switch.get(nt0, 0), sum_block=32)
    out["meta"] = np.array(repr(meta))
    out["switch_by_threads"] = np.array([[k, v] for k, v in sorted(switch.items())], np.int64)
    if not quick:
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mkl_forms.npz"), **out)
        print("\ntests/golden/mkl_forms.npz written:", meta)


if __name__ == "__main__":
    main()
