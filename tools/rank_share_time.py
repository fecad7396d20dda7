"""Time rank r's share of an N-rank frame on ONE GPU (what each rank of the multi-GPU run does
between gathers): 8-row strips r, r+N, ... of the config's frame, K frames, with one render
context or two alternating (bench.py's N>1 schedule).  Ideal = full-frame time / N.

usage: python tools/rank_share_time.py [config] [N ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import esctp1raytracer_amd as esc
from esctp1raytracer_amd import multigpu

cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
worlds = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
W, H = {"c2": (1920, 1080), "c3": (3840, 2160), "c4": (3840, 2160), "c5": (7680, 4320)}[cfg]
shadows = cfg != "c2"
scene = esc.Scene.synthetic(cfg)
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
K = 12 if cfg != "c5" else 3
for world in worlds:
    rows = multigpu.max_local_rows(H, world)
    for n_ctx in (1, 2):
        ctxs = []
        for _ in range(n_ctx):
            st = torch.cuda.Stream()
            r = esc.Renderer(0, stream=st)
            r.upload(scene)
            ctxs.append((r, st, torch.zeros(rows * W * 3, dtype=torch.uint8, device="cuda:0")))
        for timed in (False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                r, st, buf = ctxs[i % n_ctx]
                with torch.cuda.stream(st):
                    r.render_strips(cam, W, H, 0, world, out_u8=buf, shadows=shadows)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / K * 1e3
        print(f"{cfg} rank 0 of {world}: {dt:8.3f} ms/frame with {n_ctx} context(s)", flush=True)
        for r, _, _ in ctxs:
            r.close()
