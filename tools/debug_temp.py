import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from util import SMALL, small_stream, make_pair
from oracle import BATCHED
K, frames = small_stream(4)
for mode in ("plain", "adopt"):
    g, o = make_pair(SMALL, K)
    for f, (R, T, d) in enumerate(frames):
        t = torch.from_numpy(d.view(np.int16)).cuda()
        t2 = t + 0
        if mode == "plain":
            torch.cuda.synchronize()
        g.recast_depth_to_map(R, T, t2, None)
        if mode == "plain":
            g.sync()
        print(mode, f, "cur stream query", torch.cuda.current_stream().query(), "ext", list(g._ext_streams.keys()))
        del t, t2
    print(mode, "stats", g.last_frame_stats(), "count", g.count_active())
