"""CPU: the oracle's restatement of HOW the reference's matmuls and sums round (oracle/pmvo_oracle.c: mm4_elem, mm3_elem,
row_sum1 -- the code cam_project / cam_unproject / prj_loss_point are built from) against torch's own outputs recorded at
the boundary sizes where the goldens were generated: tests/golden/mkl_forms.npz, written by tools/probe_mkl_forms.py (which
also prints the full table: which expression tree each column count lands in, the switch by thread count)."""
import ast
import os

import numpy as np

import oracle
from conftest import GOLDEN


def load():
    z = np.load(os.path.join(GOLDEN, "mkl_forms.npz"), allow_pickle=False)
    return z, ast.literal_eval(str(z["meta"]))


def test_recorded_environment_is_the_one_the_rule_defaults_to():
    z, meta = load()
    assert meta["fma_min_cols"] == oracle.REF_FMA_MIN_COLS == 28445 and meta["sum_block"] == 32 and meta["threads"] == 8
    assert oracle.get_reproject_rule() == ("group", 28445)
    # MKL's switch to its threaded kernel moves with the thread count (1 thread: never) -- recorded, and the reason the
    # threshold is an option (mh_ctx_set_option "reproject_fma_min_cols", oracle.set_reproject_rule)
    sw = dict(z["switch_by_threads"].tolist())
    assert sw[1] == 0 and sw[8] == 28445 and len(set(sw.values())) > 2


def test_projection_product_forms():
    z, _ = load()
    for M in (1, 2, 5):                 # one column: its own kernel; two and more: the k-ordered chain
        got = oracle.mm4(z["mm4_M%d_A" % M], z["mm4_M%d_B" % M])
        assert np.array_equal(got, z["mm4_M%d_out" % M]), M
    # ... and the single-column form really is another one
    prev = oracle.set_reproject_rule("mid")
    try:
        assert not np.array_equal(oracle.mm4(z["mm4_M1_A"], z["mm4_M1_B"]), z["mm4_M1_out"])
    finally:
        oracle.set_reproject_rule(*prev)


def test_reprojection_product_forms():
    z, _ = load()
    for C in (1, 3, 4, 90, 28444, 28445):       # <= 3 columns chain, then separately rounded products, chain from 28445 on
        got = oracle.mm3(z["mm3_C%d_A" % C], z["mm3_C%d_B" % C])
        assert np.array_equal(got, z["mm3_C%d_out" % C]), C
    for mode, exact in (("mid", (4, 90, 28444)), ("chain", (1, 3, 28445))):
        prev = oracle.set_reproject_rule(mode)
        try:
            for C in (1, 3, 4, 90, 28444, 28445):
                same = np.array_equal(oracle.mm3(z["mm3_C%d_A" % C], z["mm3_C%d_B" % C]), z["mm3_C%d_out" % C])
                assert same == (C in exact), (mode, C)
        finally:
            oracle.set_reproject_rule(*prev)


def test_outer_sum_trailing_columns():
    z, _ = load()
    for V, C in ((24, 376), (300, 72), (20, 270)):       # C mod 32 = 24, 8, 14; 300 rows: the cascade's third level
        x, want = z["sum_V%d_C%d_x" % (V, C)], z["sum_V%d_C%d_out" % (V, C)]
        assert np.array_equal(oracle.outer_sum(x), want), (V, C)
        prev = oracle.set_sum_block(0)
        try:
            got = oracle.outer_sum(x)
        finally:
            oracle.set_sum_block(prev)
        t0 = C - C % 32
        assert np.array_equal(got[:t0], want[:t0]) and not np.array_equal(got[t0:], want[t0:]), (V, C)
