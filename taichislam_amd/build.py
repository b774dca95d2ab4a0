"""Builds libtaichislam_hip.so (the HIP/CDNA4 kernels + C-ABI) in-tree with hipcc for gfx950."""
import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libtaichislam_hip.so")
# the same library with -DTSL_TEST_HOOKS on the files that have hooks: fault injection (TSL_FAULT_NO_BDONE_WAIT) and the developer A/B switches
# (include/taichislam_hip.h, "environment switches") exist ONLY there; tests/test_pipeline_overlap_gpu.py loads it in a child process through TSL_LIB
HOOKS_LIB = os.path.join(LIBDIR, "libtaichislam_hip_testhooks.so")
HOOK_SOURCES = ("tsl_tsdf.hip", "tsl_sequential.hip")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # bit-exact parity with the CPU oracle: no FMA contraction, IEEE divide/sqrt, keep denormals
               "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
               "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) + \
        glob.glob(os.path.join(_HERE, "..", "include", "*.h"))


def source_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.hpp, *.h, the public header): stamps profiles/*_traffic.json, so that bench.py can tell whether the
    counters it replays were collected on these kernels."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_deps()):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _flags():
    return [f for f in HIPCC_FLAGS if f != "-shared"] + os.environ.get("TSL_EXTRA_FLAGS", "").split()


def _link_flags(flags):
    """the link line follows the compile flags' targets (an --offload-arch override used to compile for one target and link for another)"""
    return [f for f in flags if f.startswith("--offload-arch")] + ["-shared", "-fPIC"]


def is_stale():
    if not os.path.exists(LIB) or not os.path.exists(HOOKS_LIB):
        return True
    stamp = os.path.join(LIBDIR, "obj", "flags.txt")
    if _hipcc() is not None and (not os.path.exists(stamp) or open(stamp).read() != " ".join(_flags())):
        return True             # built with other flags (a developer -D build left behind): not the library the next normal build wants
    t = min(os.path.getmtime(LIB), os.path.getmtime(HOOKS_LIB))
    return any(os.path.getmtime(s) > t for s in _deps())


def build_library(force=False, verbose=False):
    """Compile every csrc/*.hip into lib/libtaichislam_hip.so.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    cc = _hipcc()
    if cc is None:
        if os.path.exists(LIB):
            return LIB          # GPU box without a toolchain: use the prebuilt library that travelled with the repo
        raise RuntimeError("hipcc not found and no prebuilt libtaichislam_hip.so present")
    os.makedirs(LIBDIR, exist_ok=True)
    # one object per source, compiled side by side (the seven files take ~60 s one after the other), only the stale ones; then one link
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = _flags()
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdr_t = max(os.path.getmtime(h) for h in _deps() if not h.endswith(".hip"))
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            cmd = [cc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, subprocess.Popen(cmd)))
    hook_objs = {}
    for src in sources():
        if os.path.basename(src) in HOOK_SOURCES:
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".hooks.o")
            hook_objs[os.path.basename(src)[:-4] + ".o"] = obj
            if force or not same_flags or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
                jobs.append((src, subprocess.Popen([cc] + flags + ["-DTSL_TEST_HOOKS", "-c", src, "-o", obj])))
    failed = [src for src, p in jobs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed on " + ", ".join(failed))
    open(stamp, "w").write(" ".join(flags))
    subprocess.check_call([cc] + _link_flags(flags) + objs + ["-o", LIB + ".tmp"])
    os.replace(LIB + ".tmp", LIB)
    subprocess.check_call([cc] + _link_flags(flags) + [hook_objs.get(os.path.basename(o), o) for o in objs] + ["-o", HOOKS_LIB + ".tmp"])
    os.replace(HOOKS_LIB + ".tmp", HOOKS_LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
