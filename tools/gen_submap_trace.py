"""Dev-box tool: run the REFERENCE's own orchestration (/root/reference/taichi_slam/mapping/submap_mapping.py, loaded by path, unmodified)
on recording stand-ins for the map classes and commit every call it makes into them -- arguments by value -- as
tests/golden/submap_trace.json.  tests/test_submap_trace.py replays that trace on the HIP shims on the GPU box (where the reference tree
does not exist) and compares the global map with the oracle's fusion.   python tools/gen_submap_trace.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import submap_trace as st                                                     # noqa: E402
from test_reference_callers import REF, load_reference_submap_mapping       # noqa: E402

if __name__ == "__main__":
    assert os.path.exists(REF), "the reference tree is needed to generate the trace"
    Ref = load_reference_submap_mapping(st.RecTSDF, st.RecOcto, st.RecBase)
    out = st.record(Ref)
    out["generator"] = "tools/gen_submap_trace.py: reference taichi_slam/mapping/submap_mapping.py driven through tests/submap_trace.py:drive"
    path = os.path.join(ROOT, "tests", "golden", "submap_trace.json")
    json.dump(out, open(path, "w"))
    print(f"{len(out['trace'])} calls, {len(out['sent'])} submaps sent, {os.path.getsize(path) / 1024:.0f} KiB -> {path}")
    for c in out["trace"]:
        print(f"  {c['map']:14s} {c['method']}")
