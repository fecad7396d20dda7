"""Developer tool: GPU time of ever shorter shares of the c4 frame (what does not shrink with the band)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import esctp1raytracer_amd as esc
cfg = "c4"; W, H = 3840, 2160
st = torch.cuda.Stream()
r = esc.Renderer(0, stream=st)
r.upload(esc.Scene.synthetic(cfg))
cam = esc.Camera.for_image(*esc.synthetic_view(), W, H)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda:0")
def timed(fn, n=40):
    for _ in range(3): fn()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n): fn()
    e1.record(st); st.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (1, 2, 4, 8, 16, 32, 64, 135, 270):
    t = timed(lambda: r.render_strips(cam, W, H, N // 2, N, out_u8=buf, strip_rows=8, flags=esc.ESC_RENDER_NO_COUNTERS))
    print(f"rank {N//2} of {N}: {t:.1f} us (ideal {313.0 / N:.1f})")
# contiguous bands: sky only / floor only
for name, h0, n in (("bottom 270 rows", 0, 270), ("rows 1000-1270", 1000, 272), ("top 270 rows", 1888, 272), ("8 rows at 1000", 1000, 8)):
    t = timed(lambda: r.render_rows(cam, W, H, h0, h0 + n, out_u8=buf, flags=esc.ESC_RENDER_NO_COUNTERS))
    print(f"{name}: {t:.1f} us")
