"""Developer tool: tile lists vs the three-level sweep on the icosphere twin of c2 (160x90)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import esctp1raytracer_amd as esc
import oracle_lib as ol

sc = esc.Scene.synthetic("c2", 100)
d = ol.scene_from_product(sc)
tw = ol.icosphere_twin(d, 2, smooth_normals=False)
eye, look = esc.synthetic_view()
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (160, 90)
r = esc.Renderer(0)
r.upload(ol.scene_to_product(tw))
cam = esc.Camera.for_image(eye, look, W, H)
a = r.render(cam, W, H, shadows=False)
L = r.tile_lists(1)
b = r.render(cam, W, H, shadows=False, flags=esc.ESC_RENDER_NO_TILE_LISTS)
diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
print("differing pixels", int(diff.sum()), "of", W * H)
if L is not None:
    c = L["counts"]
    print({k: v for k, v in L.items() if k != "counts"}, "max", int(c.max()), "overflowing", int((c > L["cap"]).sum()), "mean", float(c.mean()))
    hh, ww = np.nonzero(diff)
    if len(hh):
        tiles = {}
        for h, w in zip(hh, ww):
            tiles[(h // 4, w // 32)] = tiles.get((h // 4, w // 32), 0) + 1
        print("differing pixels by tile (row4, col32): count [list count]")
        for (ty, tx), n in sorted(tiles.items())[:40]:
            print("  ", ty, tx, n, int(c[ty, tx]))
        print("rows with differences:", sorted(set(hh.tolist()))[:50])
        print("example pixels", list(zip(hh[:10].tolist(), ww[:10].tolist())))
        print("lists:", a[hh[0], ww[0]], "sweep:", b[hh[0], ww[0]])
