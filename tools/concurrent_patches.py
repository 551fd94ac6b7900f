#!/usr/bin/env python3
"""Aggregate iterations/s of K independent patch optimisations replayed round-robin as hipGraphs on K streams of ONE GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
patch = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 64, 64)
steps = 60
for K in (1, 2, 3, 4, 6):
    Ts, graphs, streams = [], [], []
    for k in range(K):
        T, args = bench.make_interpolator(patch, "linear", "cuda", k)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g = T.graph_prepare()
        Ts.append(T); graphs.append(g); streams.append(s)
    torch.cuda.synchronize()
    for _ in range(5):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, s in zip(graphs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("K=%d: %.2f ms per round, %.1f patch-iterations/s" % (K, dt / steps * 1e3, K * steps / dt))
    for T in Ts:
        T.graph_finish()
    del Ts, graphs, streams
    torch.cuda.empty_cache()
