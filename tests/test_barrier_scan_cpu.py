"""tools/scan_barriers.py over the literal mode's kernels (csrc/tsl_sequential.hip compiled to gfx950 assembly here, no GPU needed): no s_barrier may be
reachable with an LDS operation of the wave still in flight.  Round 5's rare one-brick difference was exactly that -- behind stages ordered by a
wavefront-scope fence the compiler emitted a bare s_barrier in k_seq_group's bitonic network -- and the second test rebuilds that form
(-DTSL_SEQ_BITONIC -DTSL_SEQ_BITONIC_NOWAIT) to show that the scanner sees it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
needs_hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")


def _asm(tmp_path, name, extra=()):
    from taichislam_amd import build
    out = tmp_path / (name + ".s")
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-shared", "-fPIC", "-Wall", "-Wno-unused-function")]
    subprocess.check_call([HIPCC] + flags + list(extra) + ["-S", "--cuda-device-only", os.path.join(build.CSRC, "tsl_sequential.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return str(out)


@needs_hipcc
def test_no_barrier_of_the_literal_mode_is_signalled_with_lds_operations_in_flight(tmp_path):
    import scan_barriers
    found = scan_barriers.scan(_asm(tmp_path, "seq"))
    assert not found, "\n".join(f"{n}: line {l}: `{w}` (line {at}) may be pending at an s_barrier" for _, l, n, w, at in found)


@needs_hipcc
def test_the_scanner_sees_round_4s_bare_barrier(tmp_path):
    import scan_barriers
    found = scan_barriers.scan(_asm(tmp_path, "seq_r4", ["-DTSL_SEQ_BITONIC", "-DTSL_SEQ_BITONIC_NOWAIT"]))
    if not found:
        pytest.skip("this compiler waits for the LDS queue in front of that barrier by itself")
    assert all("k_seq_group" in n and w.startswith("ds_write") for _, _, n, w, _ in found), found
    # ... and with the explicit wait the same network is clean
    assert not scan_barriers.scan(_asm(tmp_path, "seq_r4_wait", ["-DTSL_SEQ_BITONIC"]))
