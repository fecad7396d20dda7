"""Developer tool: host time of one torch.distributed collective enqueue with the RCCL backend at world size 1
(what bench.py pays per frame at N > 1 besides the render call): gather 18 us, all_gather_into_tensor 16 us."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
x = torch.zeros(3110400, dtype=torch.uint8, device="cuda:0")
out = [torch.zeros_like(x)]
st = torch.cuda.Stream()
for name, fn in (("gather", lambda: dist.gather(x, out, dst=0)),
                 ("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out[0], x)),
                 ("event record+wait", lambda: st.wait_event(torch.cuda.Event()))):
    with torch.cuda.stream(st):
        for _ in range(20): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
    print(f"{name}: {1e6 * (t1 - t0) / 300:.1f} us of host time per call (world 1)")
dist.destroy_process_group()
