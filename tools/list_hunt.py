"""Developer hunt: random scenes (tests/random_scenes.py), the default frame (tile lists + light lists) against
the three-level group sweep (lists off) and, every fourth scene, the index-order sweep -- bit for bit.
python tools/list_hunt.py [n_scenes] [first_seed] [oracle_every]  (oracle_every k > 0: every k-th scene is
also rendered by the CPU oracle and compared bit for bit)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import esctp1raytracer_amd as esc
import oracle_lib as ol
from random_scenes import random_scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
oracle_every = int(sys.argv[3]) if len(sys.argv) > 3 else 0
checked = 0
r = esc.Renderer(0)
off = esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_LIGHT_LISTS
bad = lit = listed = served = cones = 0
for seed in range(first, first + n):
    d, eye, look, W, H, vfov = random_scene(seed)
    r.upload(ol.scene_to_product(d))
    cam = esc.Camera.for_image(eye, look, W, H, vfov=vfov)
    a = r.render(cam, W, H)
    st = [r.tile_lists(w) for w in range(4)]
    b = r.render(cam, W, H, flags=off)
    nd = int((a.view(np.uint32) != b.view(np.uint32)).any(axis=2).sum())
    if seed % 4 == 0:
        c = r.render(cam, W, H, flags=esc.ESC_RENDER_INDEX_ORDER)
        nd += int((a.view(np.uint32) != c.view(np.uint32)).any(axis=2).sum())
    if oracle_every and seed % oracle_every == 0:
        ref = ol.oracle_render(d, eye, look, W, H, threads=16, vfov=vfov)
        nd += int((a.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
        checked += 1
    lit += 1 if a.any() else 0
    listed += 1 if any(s is not None and s["off"] == 0 for s in st[:2]) else 0
    served += 1 if any(s is not None and s["off"] == 0 for s in st[2:]) else 0
    cones += 1 if (st[1] is not None and st[1]["cones"] > 0) else 0
    if nd:
        bad += 1
        print(f"DIFFERENCE seed={seed} pixels={nd} W={W} H={H} eye={eye} look={look} vfov={vfov}", flush=True)
    if (seed - first) % 100 == 99:
        print(f"... {seed - first + 1} scenes, {bad} differ", flush=True)
print(f"{n} scenes from seed {first}: {lit} with lit pixels, {listed} with tile lists, {served} with light lists, "
      f"{cones} with escape entries, {checked} also against the oracle, {bad} with a difference")
