"""Developer aid: every s_barrier of a gfx950 assembly listing that a wave can reach with an LDS operation of its own still in flight -- no
`s_waitcnt ... lgkmcnt(0)` on SOME path between a ds_* instruction and the barrier.  A forward data-flow over the listing's control-flow graph (labels,
branches, fall-through; loops included), one bit per program point: "an LDS operation may be pending".

Why: round 5's one-brick-in-a-few-hundred-batches difference of the literal mode.  Behind stages ordered only by a wavefront-scope fence ROCm 7.2's waitcnt
pass emitted a BARE s_barrier where __syncthreads() normally carries `s_waitcnt lgkmcnt(0)` (k_seq_group's bitonic network, reached through a loop back-edge):
another wave read LDS words whose writes were still queued.  tests/test_barrier_scan_cpu.py runs this over csrc/tsl_sequential.hip.

    hipcc --offload-arch=gfx950 -O3 <the build's flags> -S --cuda-device-only x.hip -o x.s ; python tools/scan_barriers.py x.s [more.s ...]
Prints one line per finding, exit code 1 if there is any.  (A pending ds_read whose value is only used after the barrier would be reported too; the compiler
does not produce that pattern in these kernels.)"""
import re
import sys

_LABEL = re.compile(r"^([.\w$]+):")
_BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+([.\w$]+)")


def functions(lines):
    """[(name, first line index, last line index)] of the kernels / device functions of a listing (a function ends at its s_endpgm / s_setpc / next function)"""
    out, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z[\w$]+):", l)
        if m:
            if cur:
                out.append((cur[0], cur[1], i - 1))
            cur = (m.group(1), i)
    if cur:
        out.append((cur[0], cur[1], len(lines) - 1))
    return out


def scan_function(lines, lo, hi):
    """findings [(line number, instruction that may be pending)] of one function"""
    # instructions and labels
    ins, label_at = [], {}
    for i in range(lo, hi + 1):
        t = lines[i].split(";")[0].strip()
        if not t:
            continue
        m = _LABEL.match(t)
        if m:
            label_at[m.group(1)] = len(ins)
            continue
        if t.startswith("."):
            continue
        ins.append((i, t))
    n = len(ins)
    if not n:
        return []
    succ = [[] for _ in range(n)]
    for k, (_, t) in enumerate(ins):
        m = _BRANCH.match(t)
        if t.startswith("s_endpgm") or t.startswith("s_setpc"):
            continue
        if m:
            tgt = label_at.get(m.group(2))
            if tgt is not None and tgt < n:
                succ[k].append(tgt)
            if m.group(1) == "s_branch":
                continue
        if k + 1 < n:
            succ[k].append(k + 1)
    # state BEFORE instruction k: index of a ds_* instruction that may be pending, or -1
    pend_in = [-1] * n
    seen = [False] * n
    work = [0]
    seen[0] = True
    findings = {}
    while work:
        k = work.pop()
        _, t = ins[k]
        p = pend_in[k]
        if t == "s_barrier" and p >= 0:
            findings[k] = p
        if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
            p = -1
        elif t.startswith("ds_") and not t.startswith(("ds_bpermute", "ds_permute", "ds_swizzle")):      # (the cross-lane forms use the LDS crossbar, not its memory)
            p = k
        elif t == "s_barrier":
            p = p          # (a bare barrier does not wait for the wave's own LDS queue)
        for s in succ[k]:
            if not seen[s] or (p >= 0 and pend_in[s] < 0):
                if p >= 0 and pend_in[s] < 0:
                    pend_in[s] = p
                seen[s] = True
                work.append(s)
    return [(ins[k][0] + 1, ins[p][1], ins[p][0] + 1) for k, p in sorted(findings.items())]


def scan(path, skip=("_ZN7rocprim",)):
    lines = open(path).read().split("\n")
    out = []
    for name, lo, hi in functions(lines):
        if any(name.startswith(s) for s in skip):
            continue
        for line, what, where in scan_function(lines, lo, hi):
            out.append((path, line, name, what, where))
    return out


if __name__ == "__main__":
    bad = []
    for path in sys.argv[1:]:
        bad += scan(path)
    for path, line, name, what, where in bad:
        print(f"{path}:{line} {name[:70]}: s_barrier reachable with `{what}` (line {where}) not waited for")
    sys.exit(1 if bad else 0)
