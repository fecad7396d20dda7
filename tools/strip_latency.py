import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esctp1raytracer_amd as esc
W, H = 3840, 2160
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda:0")
def timed(fn, n=40):
    for _ in range(3): fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cfg in ("c2", "c4"):
    r.upload(esc.Scene.synthetic(cfg))
    for sh in (True, False):
        for name, h0, n in (("8 sky rows at 2100", 2104, 8), ("8 rows at 1000", 1000, 8), ("8 floor rows at 100", 104, 8), ("64 rows at 1000", 1000, 64)):
            t = timed(lambda: r.render_rows(cam, W, H, h0, h0 + n, out_f32=buf, shadows=sh, flags=esc.ESC_RENDER_NO_COUNTERS))
            t2 = timed(lambda: r.render_rows(cam, W, H, h0, h0 + n, out_f32=buf, shadows=sh, flags=esc.ESC_RENDER_NO_LIGHT_LISTS | esc.ESC_RENDER_NO_TILE_LISTS | esc.ESC_RENDER_NO_COUNTERS))
            print(f"{cfg} shadows={sh} {name}: {t:.1f} us; without lists {t2:.1f} us")
