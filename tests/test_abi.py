"""CPU-only: the C-ABI library is present, loads, and exports every symbol include/mh_pmvo.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols(headers=("mh_pmvo.h", "mh_pmvo_lab.h")):
    out = set()
    for h in headers:
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        out |= set(re.findall(r"\b(mh_[a-z_0-9]+)\s*\(", hdr))
    return sorted(out)


def test_lab_switches_are_not_in_the_supported_header():
    """include/mh_pmvo.h is what an integrator binds; the A/B forms and cross-check kernels live in mh_pmvo_lab.h"""
    main, lab = set(declared_symbols(("mh_pmvo.h",))), set(declared_symbols(("mh_pmvo_lab.h",)))
    assert lab == {"mh_ctx_set_lab_option", "mh_debug_key_stats"} and not (main & lab)
    doc = open(os.path.join(ROOT, "include", "mh_pmvo.h")).read()
    for key in ("search_variant", "search_body", "taps_tile", "tap_codes"):
        assert '"%s"' % key not in doc, key


def test_header_declares_the_bound_entry_points():
    from monohair_amd import _lib

    assert set(_lib.EXPORTS) == set(declared_symbols())


def test_library_loads_and_exports_everything():
    from monohair_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert _lib.lib().mh_version() >= 100


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under monohair_amd/ (or PMVO.py) may import it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "monohair_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_no_gpu_means_loud_failure():
    import pytest
    import torch

    from monohair_amd import _lib, pmvo

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.MhError):
        pmvo.PMVO({}, {}, {}, {}, {}, device="cuda:0", image_size=[8, 8])


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md maps each C entry point to the reference code it replaces: keep it complete."""
    from monohair_amd import _lib

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert [n for n in _lib.EXPORTS if n not in doc] == []
