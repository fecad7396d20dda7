#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one frame of the workload BASELINE.json's metric is quoted on (config c4):
3840x2160, 10,000 random spheres + a 2-triangle floor + one single-triangle light, one primary
ray per pixel and one shadow ray per hit pixel, brute force, scene resident in HBM.  With N
ranks the frame is cut into 8-row strips dealt round-robin (esctp1raytracer_amd/multigpu.py),
each rank renders its strips (k_primary + k_shade) and the fp32 framebuffer is gathered to rank 0 over
RCCL and laid out as one frame there; total work is fixed, so scaling is "strong".

Rank 0 prints ONE JSON line.  `value` = rays of all ranks / wall time (max over ranks) in
Mrays/s; rays = primary (W*H) + shadow (hit pixels x lights), counted by the kernel itself.
`roofline` is the HBM view north_star asks for (algorithmic bytes / measured kernel time vs
8 TB/s) and `roofline_valu` the view that actually bounds this kernel.
`cpu_baseline` times the test oracle (oracle/rt_oracle.c, a strict-IEEE restatement of the
reference's scalar path: the reference's ISPC path cannot be built, `ispc` is not in the image)
on a bounded sample of rows of the SAME frame on all host cores, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import esctp1raytracer_amd as esc  # noqa: E402
from esctp1raytracer_amd import multigpu  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec)
FP32_VALU_PEAK_TF = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md:41
F_SPHERE, F_TRI = 19, 51   # algorithmic flop per ray-primitive test, SURVEY.md 8(d)


def source_stamp():
    """sha256 over the device sources and build flags the committed rocprofv3 numbers belong to"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "esctp1raytracer_amd", "csrc", "*"))) + \
        [os.path.join(ROOT, "Makefile")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_profile(config):
    """profiles/current.json: per-frame PMC numbers of the SHIPPED kernels (tools/summarize_prof.py
    writes it, stamped with source_stamp()).  Returns None when it belongs to other sources --
    a stale counter must never be passed off as this run's."""
    path = os.path.join(ROOT, "profiles", "current.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        prof = json.load(f)
    if prof.get("source_stamp") != source_stamp() or prof.get("config") != config:
        return None
    return prof


# Issue cost of one wave64 instruction on one SIMD with several waves resident, in cycles at the
# nominal 2.4 GHz, measured on this chip: plain VALU and v_pk_mul/add_f32 by tools/ubench/
# valu_rate.hip; v_pk_fma_f32 with one SGPR-pair operand by tools/ubench/filter_rate.hip (mode 3,
# the primary filter's loop from registers: 144.9 cycles per 24 v_pk_fma + 9 plain).  A
# v_pk_fma_f32 with three VGPR-pair operands costs 8.1 -- the filters have none in their loops.
CYC_PLAIN, CYC_PK, CYC_PK_FMA = 2.66, 4.1, 5.04
# VALU instruction mix of each hot loop (counted in the ISA, `make asm`): (pk_fma, pk mul/add, plain)
LOOP_MIX = {"k_primary": (24, 0, 9),   # per 8 spheres x 128 rays
            "k_shade": (28, 4, 5)}      # per 4 pair records (8 spheres) x 64 rays


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c4", help="c2|c3|c4|c5 (BASELINE.json configs 2-5)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--prims", type=int, default=0, help="override primitive count (0 = config's)")
    ap.add_argument("--stage", default="auto", choices=["auto", "smem", "lds", "bvh"],
                    help="auto/smem/lds are brute force (BASELINE's algorithm, the default); bvh "
                         "makes the opt-in acceleration structure the measured path")
    ap.add_argument("--no-accel", action="store_true",
                    help="skip the extra ESC_STAGE_BVH leg reported under \"accel\" (N=1 only)")
    ap.add_argument("--profile-run", action="store_true",
                    help="render default-path frames only (no index-order frame for the reference's "
                         "test count): what tools/profile.sh wants under rocprofv3, so that every "
                         "launch in the trace belongs to a timed frame's path")
    ap.add_argument("--pipelined", action="store_true",
                    help="N=1: also time the same K frames with two in flight on two streams "
                         "(information only; off by default so that a rocprofv3 trace of the "
                         "default command holds undisturbed kernel durations)")
    ap.add_argument("--gather", default="auto", choices=["auto", "f32", "u8"],
                    help="what rank 0 collects: fp32 RGB (the seam's return_image) or the PPM-"
                         "quantised bytes k_shade writes (main.cpp:676-682; 4x fewer bytes over "
                         "xGMI, SURVEY.md 8(f)2).  auto = fp32 on one GPU, u8 when there is a gather")
    ap.add_argument("--cpu-rows", type=int, default=-1,
                    help="rows of the frame the CPU baseline renders (0 = skip, -1 = as many as a "
                         "one-row-per-thread probe says fit ~12 s of CPU work)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo lets several ranks share one GPU to rehearse the N>1 path")
    ap.add_argument("--verify-rows", type=int, default=0,
                    help="N>1: rank 0 checks this many rows of the assembled frame against the "
                         "test oracle after the timed region")
    return ap.parse_args()


CONFIGS = {  # BASELINE.json configs 2-5 -> (W, H, shadows)
    "c2": (1920, 1080, False),
    "c3": (3840, 2160, True),
    "c4": (3840, 2160, True),
    "c5": (7680, 4320, True),
}


def host_cores():
    """cores this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def accel_leg(esc, r, st, cam, eye, look_at, W, H, shadows, steps, warmup, brute_frame, alg_bytes):
    """The same frame through ESC_STAGE_BVH (opt-in like the reference's --bvh):
    tree build timed apart from the render (as main.cpp:569-579 does), K frames between HIP
    events on the render stream, and EVERY fp32 value compared with the brute-force frame."""
    acc = r.build_accel(eye)
    buf = torch.zeros(H * W * 3, dtype=torch.float32, device=brute_frame.device)
    ev = []
    r.synchronize()
    t0 = 0.0
    for i in range(warmup + steps):
        if i == warmup:
            r.synchronize()
            r.reset_counters()
            r.synchronize()
            t0 = time.perf_counter()
        with torch.cuda.stream(st):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            r.render_rows(cam, W, H, 0, H, out_f32=buf, shadows=shadows, stage=esc.ESC_STAGE_BVH)
            e1.record(st)
        if i >= warmup:
            ev.append((e0, e1))
    st.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    cnt = r.counters()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    differing = int((buf.view(torch.int32) != brute_frame[:H * W * 3].view(torch.int32)).sum().item())
    # a camera that moves every frame: the per-camera constants (k_prepare_*) and the screen bins
    # are redone each time, the tree and the light bins are not (they belong to the scene)
    import math
    mv, cams = [], []
    for i in range(steps):
        ang = 2 * math.pi * i / max(steps, 1)
        e = (eye[0] + 0.3 * math.cos(ang), eye[1] + 0.1 * math.sin(ang), eye[2] + 0.3 * math.sin(ang))
        cams.append(esc.Camera.for_image(e, look_at, W, H))
    for c in cams:
        with torch.cuda.stream(st):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            r.render_rows(c, W, H, 0, H, out_f32=buf, shadows=shadows, stage=esc.ESC_STAGE_BVH)
            e1.record(st)
        mv.append((e0, e1))
    st.synchronize()
    ms_moving = sum(a.elapsed_time(b) for a, b in mv) / len(mv)
    chk = torch.zeros_like(buf)
    with torch.cuda.stream(st):
        r.render_rows(cams[-1], W, H, 0, H, out_f32=chk, shadows=shadows)  # brute force, same camera
    st.synchronize()
    moving_same = int((buf.view(torch.int32) != chk.view(torch.int32)).sum().item()) == 0
    rays = (cnt["primary_rays"] + cnt["shadow_rays"]) / steps
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"stage": "bvh", "value": rays / (ms * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_step": ms,
            "ms_per_step_wall": wall_ms,  # host clock over the same K frames, launches included
            "ms_per_step_moving_camera": ms_moving,  # new camera position every frame
            "moving_camera_last_frame_identical_to_brute_force": moving_same,
            "steps": steps, "timing": "HIP events around the whole frame: memset + k_bin_primary + k_shade<BVH> (closest hit and shading in one kernel)",
            "fp32_values_differing_from_brute_force": differing,
            "identical_to_brute_force": differing == 0,
            "tree": {k: acc[k] for k in ("tri_nodes", "tri_depth", "sph_nodes", "sph_depth")},
            "build_ms_host": acc["build_ms"],
            "leaf_tests_per_shadow_ray": cnt["anyhit_tests"] / max(cnt["shadow_rays"], 1),
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes}}


def cpu_baseline(scene, eye, look, W, H, shadows, n_rows, gpu_frame):
    """The CPU checker on evenly spaced rows of the same frame, all host cores -- its
    eight-pixel packet build (oracle/rt_oracle_fast.c, AVX2, bit-equal to rt_oracle.c: SURVEY.md
    8(d)'s cpu_fast stand-in for the host ISPC path) when the CPU has AVX2, else the scalar one.
    n_rows < 0: a probe of one row per thread sizes the sample to ~12 s.  Also a free parity spot-check of
    the GPU frame on those rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import oracle_lib as ol

    d = ol.OracleScene(ol.scene_from_product(scene))
    cores = host_cores()
    packets = ol.oracle_fast() is not None

    def spaced(n):
        return sorted({min(H - 1, int((i + 0.5) * H / n)) for i in range(n)})

    if n_rows < 0:
        probe = spaced(cores)  # rows are the unit of parallelism: one per thread
        t0 = time.perf_counter()
        ol.oracle_render_rows(d, eye, look, W, H, probe, shadows=shadows, threads=cores,
                              fast=True)
        per_row = max((time.perf_counter() - t0) / len(probe), 1e-6)
        n_rows = int(max(8, min(H, 12.0 / per_row)))
    rows = spaced(n_rows)
    t0 = time.perf_counter()
    img, cnt = ol.oracle_render_rows(d, eye, look, W, H, rows, shadows=shadows, threads=cores,
                                     fast=True)
    dt = time.perf_counter() - t0
    rays = cnt["primary_rays"] + cnt["shadow_rays"]
    same = None
    if gpu_frame is not None:
        g = gpu_frame[rows]
        same = bool(np.array_equal(g.view(np.uint32), img.view(np.uint32)))
    how = ("oracle/rt_oracle_fast.c (8-pixel AVX2 packets, bit-equal to rt_oracle.c), gcc -O3 "
           "-mavx2 -ffp-contract=off -fno-fast-math" if packets else
           "oracle/rt_oracle.c (scalar), gcc -O2 -ffp-contract=off")
    return {
        "value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "sample": f"{len(rows)} evenly spaced rows of the same {W}x{H} frame "
                  f"({rays} rays, {dt:.1f} s); {how}, row-parallel pthreads",
        "seconds": dt,
    }, same


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch through "
              f"torch.distributed.run for N>1", file=sys.stderr)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the renderer has no CPU path")
    local_rank %= torch.cuda.device_count()  # gloo rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    W0, H0, shadows = CONFIGS[a.config]
    W, H = a.width or W0, a.height or H0
    scene = esc.Scene.synthetic(a.config, a.prims)
    info = scene.info()
    eye, look = esc.synthetic_view()
    cam = esc.Camera.for_image(eye, look, W, H)
    stage = {"auto": esc.ESC_STAGE_AUTO, "smem": esc.ESC_STAGE_SMEM, "lds": esc.ESC_STAGE_LDS,
             "bvh": esc.ESC_STAGE_BVH}[a.stage]

    st = torch.cuda.Stream(device=dev)
    r = esc.Renderer(local_rank, stream=st)
    r.upload(scene)  # scene resident in HBM before any timed region

    S = multigpu.STRIP_ROWS
    my_rows = multigpu.local_rows(H, rank, world, S)
    max_rows = multigpu.max_local_rows(H, world, S)
    use_u8 = a.gather == "u8" or (a.gather == "auto" and world > 1)
    ch_dtype = torch.uint8 if use_u8 else torch.float32
    n_local = max_rows * W * 3
    # N>1: two strip buffers so the gather of frame i (second stream) overlaps the render of
    # frame i+1; rank 0 assembles on that second stream through its own context
    n_buf = 2 if world > 1 else 1
    local = [torch.zeros(n_local, dtype=ch_dtype, device=dev) for _ in range(n_buf)]
    gathered = frame = r2 = st2 = None
    if world > 1:
        st2 = torch.cuda.Stream(device=dev)
        if rank == 0:
            gathered = [torch.zeros(world, n_local, dtype=ch_dtype, device=dev) for _ in range(n_buf)]
            frame = torch.zeros(H * W * 3, dtype=ch_dtype, device=dev)
            r2 = esc.Renderer(local_rank, stream=st2)
    gather_lists = [list(g.unbind(0)) for g in gathered] if gathered is not None else None
    buf_free = [None] * n_buf  # event: the gather that last read local[b] has finished
    # N>1: consecutive frames also alternate between TWO render contexts / streams.  A rank's
    # share of the frame is a small grid (N=8: 2,040 + 4,080 workgroups) whose last workgroups
    # leave most of the chip idle; with the next frame already queued on the other stream those
    # CUs have work (measured on one GPU rendering rank 0's share of 8: 2.89 -> 2.35 ms per frame,
    # ideal 18.4/8 = 2.30).  N=1 keeps one stream so kernel.avg_ms is an undisturbed duration.
    renderers, streams = [r], [st]
    if world > 1:
        stb = torch.cuda.Stream(device=dev)
        rb = esc.Renderer(local_rank, stream=stb)
        rb.upload(scene)
        renderers.append(rb)
        streams.append(stb)

    events = []
    # every event a step needs exists before the timed region (at N = 8 a rank's share of the
    # frame is ~0.1 ms: the host side of a step must not be the slower half)
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            for _ in range(a.steps + a.warmup + 8)]
    free_ev = [torch.cuda.Event() for _ in range(n_buf)]
    pool_next = [0]

    def step(i, timed):
        b = i % n_buf
        rr, ss = renderers[b % len(renderers)], streams[b % len(streams)]
        with torch.cuda.stream(ss):
            if buf_free[b] is not None:
                ss.wait_event(buf_free[b])
            if pool_next[0] < len(pool):
                e0, e1 = pool[pool_next[0]]
                pool_next[0] += 1
            else:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
            e0.record(ss)
            rr.render_strips(cam, W, H, rank, world, out_f32=None if use_u8 else local[b],
                             out_u8=local[b] if use_u8 else None, strip_rows=S, shadows=shadows,
                             stage=stage)
            e1.record(ss)
            if timed:
                events.append((e0, e1))
        if world > 1:
            with torch.cuda.stream(st2):
                st2.wait_event(e1)  # the collective orders itself after the current stream
                multigpu.gather_to_root(local[b], rank, world, gathered[b] if rank == 0 else None,
                                        gather_list=gather_lists[b] if rank == 0 else None)
                if rank == 0:
                    r2.assemble_strips(gathered[b], world, n_local * local[b].element_size(), W, H,
                                       frame, strip_rows=S,
                                       bytes_per_pixel=3 * local[b].element_size())
                ev = free_ev[b]  # its previous record has been waited on by this step's render
                ev.record(st2)
                buf_free[b] = ev

    def fence():
        for ss in streams:
            ss.synchronize()
        if st2 is not None:
            st2.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i, False)
    fence()
    for rr in renderers:
        rr.reset_counters()
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    fence()
    elapsed = time.perf_counter() - t0
    cnt = renderers[0].counters()  # rays of exactly the K timed frames
    for rr in renderers[1:]:
        for k, v in rr.counters().items():
            cnt[k] += v

    # per-kernel durations (HIP events recorded by the library between the frame's kernels):
    # a few extra frames outside the timed region, one at a time
    kernel_split = None
    if world == 1 and a.stage != "bvh":
        acc = [0.0, 0.0]
        n_split = 3
        for _ in range(n_split):
            with torch.cuda.stream(st):
                r.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else local[0],
                                out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                                stage=stage, flags=esc.ESC_RENDER_TIME_KERNELS)
            ms = r.last_kernel_ms()
            acc[0] += ms[0]
            acc[1] += ms[1]
        kernel_split = {"k_primary_ms": acc[0] / n_split, "k_shade_ms": acc[1] / n_split,
                        "note": "k_shade_ms = everything after k_primary: the fused k_shade, or "
                                "k_shadow_setup + k_anyhit_segment x segments + k_shade_finish"}

    # the reference's any-hit count: one frame in index order (the default sweeps long sphere lists
    # in another order for the last light and so executes fewer tests; the image is the same)
    anyhit_index_order = None
    index_order_ms = None
    if world == 1 and a.stage != "bvh" and not a.profile_run:
        r.synchronize()
        r.reset_counters()
        ei0, ei1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(4):  # the first such frame allocates the queue form's scratch: best of 4
            with torch.cuda.stream(st):
                r.reset_counters()
                ei0.record(st)
                r.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else local[0],
                                out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                                stage=stage, flags=esc.ESC_RENDER_INDEX_ORDER)
                ei1.record(st)
            r.synchronize()
            ms_i = ei0.elapsed_time(ei1)
            index_order_ms = ms_i if index_order_ms is None else min(index_order_ms, ms_i)
        anyhit_index_order = r.counters()["anyhit_tests"]
        r.reset_counters()
    # one un-pipelined frame: launch -> complete frame resident on rank 0
    fence()
    t1 = time.perf_counter()
    step(0, False)
    fence()
    frame_latency_ms = (time.perf_counter() - t1) * 1e3
    r.synchronize()

    # N=1, for information only (the headline stays one frame at a time so that kernel.avg_ms is an
    # undisturbed duration): the same K frames with two in flight on two streams, which fills the
    # tail of one frame's grids with the next frame's work
    pipelined = None
    if world == 1 and a.stage != "bvh" and a.pipelined:
        stb = torch.cuda.Stream(device=dev)
        rb = esc.Renderer(local_rank, stream=stb)
        rb.upload(scene)
        other = torch.zeros_like(local[0])
        pair = [(r, st, local[0]), (rb, stb, other)]
        for timed in (False, True):
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for i in range(a.steps):
                rr, ss, out = pair[i % 2]
                with torch.cuda.stream(ss):
                    rr.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else out,
                                     out_u8=out if use_u8 else None, strip_rows=S, shadows=shadows,
                                     stage=stage)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp
        same = bool(torch.equal(local[0], other))
        pipelined = {"frames_in_flight": 2, "ms_per_step": dtp / a.steps * 1e3,
                     "frames_identical": same}
        rb.close()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    c = torch.tensor([cnt["primary_rays"], cnt["shadow_rays"], cnt["anyhit_tests"],
                      cnt["hit_pixels"], cnt["anyhit_lane_tests"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    primary, shadow, anyhit, hits, lane_tests = (float(x) for x in c.tolist())
    kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in events) / max(len(events), 1)
    # N>1 keeps two frames in flight, so an event pair also spans time given to the other frame;
    # the rooflines there use this rank's throughput time per frame instead
    roof_ms = kernel_ms if world == 1 else elapsed / a.steps * 1e3

    if rank == 0:
        rays = primary + shadow
        n_tri, n_sph = info["n_triangles"], info["n_spheres"]
        # ---- roofline of the frame kernels (k_primary + k_shade) on THIS rank
        frame_bytes = my_rows * W * 3 * local[0].element_size()
        scene_bytes = n_sph * 32 + n_tri * 112 + (info["n_geometry"] + n_sph) * 64
        alg_bytes = frame_bytes + scene_bytes
        gbs = alg_bytes / (roof_ms * 1e-3) / 1e9
        prof = committed_profile(a.config) if (world == 1 and not a.prims and not a.width
                                               and a.stage == "auto") else None
        traffic = prof["hbm_bytes_per_frame"] if prof else None
        share = my_rows / H
        closest_flop = (primary / a.steps) * share * (n_tri * F_TRI + n_sph * F_SPHERE)
        f_any = (n_tri * F_TRI + n_sph * F_SPHERE) / max(n_tri + n_sph, 1)
        ref_anyhit = anyhit_index_order if anyhit_index_order is not None else anyhit / a.steps
        anyhit_flop = ref_anyhit * share * f_any
        tf = (closest_flop + anyhit_flop) / (roof_ms * 1e-3) / 1e12
        out = {
            "metric": "Mrays/sec + frame ms, 3840x2160 / 10k spheres, at 1/2/4/8 MI355X",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "frame_latency_ms": frame_latency_ms,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{a.config}: {W}x{H}, {n_sph} spheres + {n_tri} triangles, "
                            f"{info['n_lights']} light, 1 primary ray/pixel + "
                            f"{'1 shadow ray per hit pixel' if shadows else 'no shadow rays'}, "
                            f"{'bounding-volume tree (opt-in)' if a.stage == 'bvh' else 'brute force'}",
                "sweep": ("every (ray, primitive) pair is decided by the reference arithmetic or by a proven "
                          "conservative filter; from 64 primitives up the filters run on bounding spheres / "
                          "normal cones of spatial groups of 8, 64-128 and 512-1,024 primitives first "
                          "(DESIGN.md 3.6-3.7).  index_order_frame_ms = the same frame swept linearly in the "
                          "reference's index order (ESC_RENDER_INDEX_ORDER), best of four such frames, same run"),
                "index_order_frame_ms": index_order_ms,
                "scene": "esc_scene_synthetic (SURVEY.md 8(d)): splitmix64 seeds 0xC2..0xC4",
                "partition": f"{S}-row strips round-robin over {world} rank(s)",
                "gather": ("none (1 GPU)" if world == 1 else
                           f"{'RCCL' if a.backend == 'nccl' else 'gloo'} gather of "
                           f"{'u8' if use_u8 else 'fp32'} RGB strips to rank 0 + k_assemble_strips, "
                           f"on a second stream overlapping the next frame's render; consecutive "
                           f"frames alternate between two render streams"),
                "stage": a.stage,
                "rays_per_frame": rays / a.steps,
                "primary_rays_per_frame": primary / a.steps,
                "shadow_rays_per_frame": shadow / a.steps,
                "hit_pixels_per_frame": hits / a.steps,
                "closest_hit_tests_per_frame": primary / a.steps * (n_tri + n_sph),
                "anyhit_tests_per_frame": anyhit / a.steps,
                "anyhit_tests_per_frame_index_order": anyhit_index_order,
                "shadow_lane_efficiency": (anyhit / lane_tests) if lane_tests else None,
            },
            "kernel": {"name": "one frame = k_primary + the shading kernels (fused k_shade for short "
                               "primitive lists; k_shadow_setup + k_anyhit_segment per segment + "
                               "k_shade_finish per light otherwise), back to back on one stream"
                               + ("" if world == 1 else "; N>1: two frames are in flight on two "
                                  "streams, so this duration includes time shared with the other frame"),
                       "avg_ms": kernel_ms,
                       "launches_timed": len(events), "rank": 0, "rows": my_rows},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_source": (prof["source"] if prof else
                                            "null: profiles/current.json is missing or was "
                                            "measured on other sources (stamp mismatch)"),
                         "traffic_note": "algorithmic = fp32 framebuffer + scene tables (SURVEY.md "
                                         "8(d)); measured traffic (rocprofv3 --pmc FETCH_SIZE / "
                                         "WRITE_SIZE passes, fetch doubled per the gfx950 note) also "
                                         "holds the hit planes k_primary hands over and the shadow-ray "
                                         "queue; the kernels are VALU-bound, see roofline_valu"},
            "roofline_valu": {"bound": "valu_fp32", "achieved": tf, "peak": FP32_VALU_PEAK_TF,
                              "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TF,
                              "note": "algorithmic flop (19/sphere test, 51/triangle test) of the "
                                      "tests the REFERENCE executes / kernel time.  Not an "
                                      "executed-op rate: the kernels hoist per-primitive work and "
                                      "run a 4- / 8-op FMA filter instead of the 7- / 16-op test for "
                                      "all but the candidate pairs, so this can exceed 1; the "
                                      "executed view is `kernels`"},
        }
        if kernel_split is not None:
            out["kernel_split"] = kernel_split
            # executed-instruction view per kernel: VALU wave-instructions (committed rocprofv3
            # SQ_INSTS_VALU, stamped) x measured issue cost / (kernel time x SIMDs x clock)
            ks = {}
            for name, ms_key in (("k_primary", "k_primary_ms"), ("k_shade", "k_shade_ms")):
                ent = {"ms": kernel_split[ms_key]}
                if prof and name in prof.get("valu_insts_per_frame", {}):
                    insts = prof["valu_insts_per_frame"][name]
                    nf, npk, npl = LOOP_MIX[name]
                    cyc = (nf * CYC_PK_FMA + npk * CYC_PK + npl * CYC_PLAIN) / (nf + npk + npl)
                    clock_ghz = prof.get("clock_ghz", 2.4)
                    avail = ent["ms"] * 1e-3 * clock_ghz * 1e9 * 1024  # SIMD-cycles
                    # executed VALU instructions per SIMD-cycle against what the pipe can issue:
                    # between 1 / 5.04 (nothing but v_pk_fma_f32) and 1 / 2.66 (nothing but plain
                    # VALU); `issue_frac` prices every instruction at the filter bodies' mix and
                    # can exceed 1 where the sweeps' bookkeeping (cheaper plain VALU) dominates
                    ent.update({"valu_insts": insts, "clock_ghz": clock_ghz,
                                "valu_insts_per_simd_cycle": insts / avail,
                                "issue_ceiling_per_simd_cycle": [1 / CYC_PK_FMA, 1 / CYC_PLAIN],
                                "filter_body_mix_pkfma_pk_plain": [nf, npk, npl],
                                "cycles_per_inst_at_body_mix": cyc,
                                "issue_frac": insts * cyc / avail})
                ks[name] = ent
            out["kernels"] = ks
        if pipelined is not None:
            pipelined["value"] = rays / a.steps / (pipelined["ms_per_step"] * 1e-3) / 1e6
            out["pipelined"] = pipelined
        if world == 1 and a.cpu_rows != 0:
            gpu_frame = None
            if not use_u8:
                gpu_frame = local[0][:H * W * 3].cpu().numpy().reshape(H, W, 3)
            cb, same = cpu_baseline(scene, eye, look, W, H, shadows, a.cpu_rows, gpu_frame)
            out["cpu_baseline"] = cb
            out["gpu_vs_cpu"] = out["value"] / cb["value"]
            out["parity_sample_rows_bit_exact"] = same
        if world == 1 and a.stage != "bvh" and not a.no_accel and not use_u8:
            out["accel"] = accel_leg(esc, r, st, cam, eye, look, W, H, shadows, a.steps, a.warmup,
                                     local[0], alg_bytes)
        if world > 1 and a.verify_rows > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import numpy as np

            import oracle_lib as ol
            rows = sorted({min(H - 1, int((i + 0.5) * H / a.verify_rows)) for i in range(a.verify_rows)})
            ref, _ = ol.oracle_render_rows(ol.scene_from_product(scene), eye, look, W, H, rows,
                                           shadows=shadows, threads=host_cores())
            got = frame.view(H, W, 3)[rows].cpu().numpy()
            if use_u8:  # the PPM bytes of the same rows (main.cpp:676-682 clamp + truncate)
                same = np.array_equal(got, ol.oracle_quantise(ref))
            else:
                same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
            out["assembled_frame_rows_bit_exact"] = bool(same)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
