"""CPU-side checks of the boundary: the C-ABI library loads here (no GPU) and exports every symbol that
include/taichislam_hip.h declares; without a device the product path fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "taichislam_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from taichislam_amd import _lib
    names = _declared_symbols()
    assert len(names) > 50
    L = ctypes.CDLL(_lib.library_path()) if os.path.exists(_lib.library_path()) else _lib.lib()
    L = _lib.lib()
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"declared but not exported: {missing}"
    unbound = [n for n in names if n not in _lib.SIGNATURES]
    assert not unbound, f"declared but not bound by the ctypes shim: {unbound}"
    assert L.tsl_version().startswith(b"taichislam_hip")


def test_every_backend_option_is_documented_in_the_header():
    """tsl_tsdf_set_option / tsl_tsdf_get_option take their names as strings: every name the library compares against must be described in include/taichislam_hip.h."""
    hdr = open(os.path.join(ROOT, "include", "taichislam_hip.h")).read()
    names = set()
    for f in ("tsl_tsdf.hip", "tsl_esdf.hip", "tsl_octo.hip"):
        names |= set(re.findall(r'strcmp\(name, "([a-z_0-9]+)"\)', open(os.path.join(ROOT, "taichislam_amd", "csrc", f)).read()))
    assert len(names) > 20
    missing = sorted(n for n in names if f'"{n}"' not in hdr)
    assert not missing, f"options without a line in the header: {missing}"


def test_no_silent_cpu_fallback():
    from taichislam_amd import _lib
    from taichislam_amd.mapping import DenseTSDF
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.TslError, match="no HIP device"):
        DenseTSDF()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "taichislam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "liboracle" not in src and "tsl_oracle" not in src, f"{f} references the oracle"


def test_colormap_matches_matplotlib_jet_shape():
    from taichislam_amd.mapping.mapping_common import jet_colormap
    cm = jet_colormap()
    assert cm.shape == (1024, 3) and cm.min() >= 0 and cm.max() <= 1
    assert abs(cm[0, 2] - 0.5) < 1e-6 and cm[0, 0] == 0          # jet(0) = (0, 0, 0.5)
    assert abs(cm[1023, 0] - 0.5) < 1e-2 and cm[1023, 2] == 0    # jet(~1) = (0.5, 0, 0)


def test_comm_and_merge_argument_errors_are_reported_not_crashed():
    """tsl_comm_* / tsl_tsdf_allreduce_merge reject bad arguments before anything touches RCCL or a device (the multi-rank path cannot be
    exercised on the one-GPU boxes, so its argument checking at least is)."""
    import ctypes as C
    from taichislam_amd import _lib
    L = _lib.lib()
    h = C.c_void_p()
    buf = C.create_string_buffer(128)
    for nranks, rank in ((0, 0), (2, 2), (2, -1)):
        assert L.tsl_comm_create(buf, nranks, rank, 0, C.byref(h)) == -1 and not h.value
        assert b"comm_create" in L.tsl_last_error()
    assert L.tsl_comm_create(None, 1, 0, 0, C.byref(h)) == -1
    assert L.tsl_comm_unique_id(None) == -1
    assert L.tsl_comm_handle(None) is None
    L.tsl_comm_destroy(None)                                   # a no-op
    n = C.c_int64()
    assert L.tsl_tsdf_allreduce_merge(None, None, None, C.byref(n)) == -1 and b"allreduce_merge" in L.tsl_last_error()
    assert L.tsl_tsdf_merge_begin(None, None, None, 0) == -1
