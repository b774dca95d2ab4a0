import sys, os, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from util import SMALL, make_pair, small_stream, lin
from oracle import BATCHED
K, frames = small_stream(3)
g, o = make_pair(SMALL, K)
for a in sys.argv[1:]:
    k, v = a.split("="); g.set_option(k, int(v))
hist = []
for R, T, d in frames:
    g.recast_depth_to_map(R, T, d, None); o.integrate_depth(R, T, d, mode=BATCHED)
    e = o.export_sparse()
    hist.append({k: (float(t), float(w)) for k, t, w in zip(lin(e["indices"]).tolist(), np.asarray(e["TSDF"]).tolist(), np.asarray(e["W_TSDF"]).tolist())})
ge = g.export_submap()
gd = {k: (float(t), float(w), i) for k, t, w, i in zip(lin(ge["indices"]).tolist(), np.asarray(ge["TSDF"]).tolist(), np.asarray(ge["W_TSDF"]).tolist(), ge["indices"].tolist())}
bad = [k for k in gd if k in hist[2] and gd[k][:2] != hist[2][k]]
print("mismatches", len(bad))
kinds = collections.Counter()
for k in bad:
    gv = gd[k][:2]
    kinds["equals oracle after frame 1" if hist[1].get(k) == gv else ("equals oracle after frame 0" if hist[0].get(k) == gv else "other")] += 1
print(kinds)
for k in bad[:6]:
    print(gd[k], "oracle f0/f1/f2:", hist[0].get(k), hist[1].get(k), hist[2].get(k))
loc = collections.Counter()
for k in bad:
    i = np.array(gd[k][2]); b = tuple(((i + 128) // 16).tolist()); l = ((i + 128) % 16).tolist()
    loc[(b, l[0])] += 1
print("mismatch count by (brick, local x):", sorted(loc.items()))
