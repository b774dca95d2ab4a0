"""Developer aid: every s_barrier of a gfx950 assembly listing whose wave may still have LDS operations in flight when it signals (no
`s_waitcnt ... lgkmcnt(0)` between the wave's last ds_* instruction and the barrier, walking straight back through the listing).  Round 5: ROCm 7.2's
waitcnt pass emitted a bare s_barrier behind a stage that was ordered by a wavefront-scope fence only (k_seq_group's bitonic network).
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only x.hip -o x.s ; python tools/scan_barriers.py x.s"""
import re
import sys

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    fn = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            fn = m.group(1)
        if l.strip() != "s_barrier" or (fn or "").startswith("_ZN7rocprim"):
            continue
        k, verdict = i - 1, "ok (start of function)"
        while k > 0:
            t = lines[k].strip()
            k -= 1
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if re.match(r"^_Z\w+:", t):
                break
            if t.endswith(":"):
                continue                      # a label: other paths join here -- keep walking the fall-through path
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                verdict = None
                break
            if t == "s_barrier":
                verdict = None                # nothing LDS-bound since the previous barrier on this path
                break
            if t.startswith("ds_"):
                verdict = f"LDS op in flight? line {k + 2}: {t}"
                break
        if verdict and verdict.startswith("LDS"):
            print(f"{path}:{i + 1} {fn[:70]}: {verdict}")
