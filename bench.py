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
The timed path is a CULLED sweep -- per-tile and per-light-cell primitive lists from projected
conservative bounds, spatial groups behind them (csrc/rt_lists.h) -- in front of
the same filters and reference arithmetic, and `config.workload` says so; `linear` is the same frame with every (ray, primitive) pair visited in
the reference's index order (ESC_RENDER_INDEX_ORDER) -- BASELINE's "brute-force intersect" point.
`roofline` is the HBM view north_star asks for (algorithmic bytes / measured kernel time vs
8 TB/s); `valu` holds what actually bounds these kernels, from the committed rocprofv3 counters
(VALU busy, wait shares, instructions per wave), for both paths.
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
N_SIMD, N_XCD = 1024, 8   # 256 CUs x 4 SIMDs in 8 XCDs (MI355X_MICROARCH.md)


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
    """profiles/current.json: per-frame PMC numbers of the SHIPPED kernels, culled and linear paths
    (tools/summarize_prof.py writes it, stamped with source_stamp()).  Returns None when it belongs to
    other sources -- a stale counter must never be passed off as this run's."""
    path = os.path.join(ROOT, "profiles", "current.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        prof = json.load(f)
    if prof.get("source_stamp") != source_stamp() or prof.get("config") != config or "paths" not in prof:
        return None
    return prof


def counter_view(kern):
    """One kernel class of profiles/current.json -> the numbers that say what bounds it.  SQ counters
    are summed over the chip; SQ_ACTIVE_INST_VALU and the SQ_WAIT_* / SQ_WAVE_CYCLES counters tick in
    quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, counter section)."""
    g = kern.get
    out = {}
    if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
        out["valu_busy"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * g("GRBM_GUI_ACTIVE") / N_XCD)
    if g("SQ_WAVE_CYCLES"):
        if g("SQ_WAIT_ANY") is not None:
            out["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_WAIT_INST_ANY") is not None:
            out["wait_inst_any_share_of_wave_cycles"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_WAVES"):
        for c, k in (("SQ_INSTS_VALU", "valu_insts_per_wave"), ("SQ_INSTS_SALU", "salu_insts_per_wave"),
                     ("SQ_INSTS_SMEM", "smem_insts_per_wave")):
            if g(c) is not None:
                out[k] = g(c) / g("SQ_WAVES")
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"):
        if g(c) is not None:
            out[c.lower() + "_per_frame"] = g(c)
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="c4", help="c2|c3|c4|c5 (BASELINE.json configs 2-5)")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--prims", type=int, default=0, help="override primitive count (0 = config's)")
    ap.add_argument("--stage", default="auto", choices=["auto", "smem", "lds", "bvh"],
                    help="auto/smem/lds are brute force (BASELINE's algorithm, the default); bvh "
                         "makes the opt-in acceleration structure the measured path")
    ap.add_argument("--path", default="culled", choices=["culled", "linear"],
                    help="culled (default): tile lists / light lists / spatial groups decide which "
                         "primitives a ray's filters and reference arithmetic run on; linear: "
                         "ESC_RENDER_INDEX_ORDER, every pair in the reference's index order, is the "
                         "TIMED path (what tools/profile.sh wraps for the `linear` counters)")
    ap.add_argument("--linear-steps", type=int, default=-1,
                    help="frames of the `linear` leg (default: min(steps, 5); 0 = skip)")
    ap.add_argument("--graph", action="store_true",
                    help="timed steps replay a recorded frame (esc_frame_record: one hipGraphLaunch "
                         "per frame) instead of calling esc_render_strips (parameter block + kernel "
                         "launches per frame).  Off by default: a frame is ONE kernel, the plain call "
                         "costs ~6 us of host time, and graph replays run ~20 us per frame slower on "
                         "the GPU back to back (`host` reports both)")
    ap.add_argument("--no-accel", action="store_true",
                    help="skip the extra ESC_STAGE_BVH leg reported under \"accel\" (N=1 only)")
    ap.add_argument("--count-every-frame", action="store_true",
                    help="keep the library's ray counters on in the timed frames.  By default the timed "
                         "frames run with ESC_RENDER_NO_COUNTERS and the rays are counted on ONE identical "
                         "frame before the timed region: the counters are instrumentation the reference does "
                         "not have (two barriers + atomics per workgroup: 9 %% of a c4 frame), and every "
                         "frame of a fixed scene, camera and option set counts the same rays")
    ap.add_argument("--profile-run", action="store_true",
                    help="render whole frames of the timed path only (no index-order frame for the "
                         "reference's test count, no rank-share launches): what tools/profile.sh wants "
                         "under rocprofv3, so that every launch in the trace belongs to a timed frame's "
                         "path and the trace's average duration is a frame's")
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


def accel_leg(esc, r, st, cam, eye, look_at, W, H, shadows, steps, warmup, brute_frame, alg_bytes,
              count_every_frame=False):
    """The same frame through ESC_STAGE_BVH (opt-in like the reference's --bvh):
    tree build timed apart from the render (as main.cpp:569-579 does), K frames between HIP
    events on the render stream, and EVERY fp32 value compared with the brute-force frame."""
    acc = r.build_accel(eye)
    buf = torch.zeros(H * W * 3, dtype=torch.float32, device=brute_frame.device)
    ev = []
    r.synchronize()
    t0 = 0.0
    tree_flags = esc.ESC_RENDER_BVH_HEURISTIC_PADS
    timed_flags = tree_flags | (0 if count_every_frame else esc.ESC_RENDER_NO_COUNTERS)
    frame_cnt = None
    for i in range(warmup + steps):
        if i == warmup:
            r.synchronize()
            r.reset_counters()
            if not count_every_frame:  # one counted frame, then K timed ones without the counters
                with torch.cuda.stream(st):
                    r.render_rows(cam, W, H, 0, H, out_f32=buf, shadows=shadows, stage=esc.ESC_STAGE_BVH,
                                  flags=tree_flags)
                r.synchronize()
                frame_cnt = r.counters()
            r.synchronize()
            t0 = time.perf_counter()
        with torch.cuda.stream(st):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            r.render_rows(cam, W, H, 0, H, out_f32=buf, shadows=shadows, stage=esc.ESC_STAGE_BVH,
                          flags=timed_flags)
            e1.record(st)
        if i >= warmup:
            ev.append((e0, e1))
    st.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    cnt = r.counters() if frame_cnt is None else {k: v * steps for k, v in frame_cnt.items()}
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
            r.render_rows(c, W, H, 0, H, out_f32=buf, shadows=shadows, stage=esc.ESC_STAGE_BVH,
                          flags=timed_flags)
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
    return {"stage": "bvh", "bounds": "the tree and its bins, ESC_RENDER_BVH_HEURISTIC_PADS: proven box pads for "
                                      "spheres (and the <= 4 triangles of c2-c4 are tested directly); a triangle "
                                      "MESH (c5) gets heuristic pads -- without the flag ESC_STAGE_BVH serves meshes "
                                      "through the default path's proven lists and groups",
            "value": rays / (ms * 1e-3) / 1e6, "unit": "Mrays/s", "ms_per_step": ms,
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
    path_flags = esc.ESC_RENDER_INDEX_ORDER if a.path == "linear" else 0
    # the TIMED frames: the same, without the ray counters (see --count-every-frame)
    timed_flags = path_flags | (0 if a.count_every_frame else esc.ESC_RENDER_NO_COUNTERS)

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

    # --graph: the frame of each (renderer, buffer) pair recorded once (HIP graph): a timed step is
    # then ONE host call
    recorded = [None] * n_buf
    if a.graph:
        for b in range(n_buf):
            with torch.cuda.stream(streams[b % len(streams)]):
                recorded[b] = renderers[b % len(renderers)].record_strips(
                    cam, W, H, rank, world, out_f32=None if use_u8 else local[b],
                    out_u8=local[b] if use_u8 else None, strip_rows=S, shadows=shadows, stage=stage,
                    flags=timed_flags)
        for rr in renderers:
            rr.synchronize()

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
        # (no `with torch.cuda.stream(...)` here: the library launches on the renderer's own stream and
        # the events name theirs; the context manager alone cost ~10 us of host time per step)
        if buf_free[b] is not None:
            ss.wait_event(buf_free[b])
        if pool_next[0] < len(pool):
            e0, e1 = pool[pool_next[0]]
            pool_next[0] += 1
        else:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
        e0.record(ss)
        if recorded[b] is not None:
            recorded[b].launch()
        else:
            rr.render_strips(cam, W, H, rank, world, out_f32=None if use_u8 else local[b],
                             out_u8=local[b] if use_u8 else None, strip_rows=S, shadows=shadows,
                             stage=stage, flags=timed_flags)
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
    frame_cnt = None
    if not a.count_every_frame:
        # the rays of ONE frame of this rank, counted outside the timed region on the frame the timed
        # steps render (same scene, camera, strips and options; only the counters differ)
        with torch.cuda.stream(st):
            r.render_strips(cam, W, H, rank, world, out_f32=None if use_u8 else local[0],
                            out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                            stage=stage, flags=path_flags)
        r.synchronize()
        frame_cnt = r.counters()
        r.reset_counters()
    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i, True)
    fence()
    elapsed = time.perf_counter() - t0
    if frame_cnt is not None:
        cnt = {k: v * a.steps for k, v in frame_cnt.items()}  # K identical frames
    else:
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
                                stage=stage, flags=esc.ESC_RENDER_TIME_KERNELS | path_flags)
            ms = r.last_kernel_ms()
            acc[0] += ms[0]
            acc[1] += ms[1]
        kernel_split = {"k_primary_ms": acc[0] / n_split, "k_shade_ms": acc[1] / n_split,
                        "note": "the TWO-kernel variant of the frame (ESC_RENDER_TIME_KERNELS records an "
                                "event between the halves, so it renders with k_primary + k_shade instead "
                                "of the one-kernel k_frame the timed steps use: a few per cent slower in "
                                "total); k_shade_ms = everything after k_primary"}

    # `linear`: the same frame with every (ray, primitive) pair visited in the reference's index
    # order (ESC_RENDER_INDEX_ORDER: no groups, no re-ordered last light) -- BASELINE's "brute-force
    # intersect" point, and the reference's own any-hit count.  A few frames between HIP events on
    # the render stream, after the timed region; the first one (untimed) allocates the queue form's
    # scratch.
    linear = None
    anyhit_index_order = None
    n_lin = a.linear_steps if a.linear_steps >= 0 else min(a.steps, 5)
    if world == 1 and a.stage != "bvh" and not a.profile_run and a.path == "culled" and n_lin > 0:
        lin_ev = []
        for i in range(n_lin + 1):
            with torch.cuda.stream(st):
                if i == 1:
                    r.reset_counters()
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                r.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else local[0],
                                out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                                stage=stage, flags=esc.ESC_RENDER_INDEX_ORDER)
                e1.record(st)
            if i:
                lin_ev.append((e0, e1))
        r.synchronize()
        lc = r.counters()
        lin_ms = sum(x.elapsed_time(y) for x, y in lin_ev) / len(lin_ev)
        with torch.cuda.stream(st):
            r.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else local[0],
                            out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                            stage=stage, flags=esc.ESC_RENDER_INDEX_ORDER | esc.ESC_RENDER_TIME_KERNELS)
        lsplit = r.last_kernel_ms()
        anyhit_index_order = lc["anyhit_tests"] / n_lin
        linear = {"flag": "ESC_RENDER_INDEX_ORDER", "steps": n_lin, "ms_per_step": lin_ms,
                  "rays_per_frame": (lc["primary_rays"] + lc["shadow_rays"]) / n_lin,
                  "value": (lc["primary_rays"] + lc["shadow_rays"]) / n_lin / (lin_ms * 1e-3) / 1e6,
                  "unit": "Mrays/s",
                  "anyhit_tests_per_frame": anyhit_index_order,
                  "shadow_lane_efficiency": (lc["anyhit_tests"] / lc["anyhit_lane_tests"]
                                             if lc["anyhit_lane_tests"] else None),
                  "kernel_split": {"k_primary_ms": lsplit[0], "shading_ms": lsplit[1]}}
        r.reset_counters()
        # the timed path's frame back in local[0] (parity spot-check and the BVH comparison read it)
        with torch.cuda.stream(st):
            r.render_strips(cam, W, H, 0, 1, out_f32=None if use_u8 else local[0],
                            out_u8=local[0] if use_u8 else None, strip_rows=S, shadows=shadows,
                            stage=stage, flags=path_flags)
        r.synchronize()
    # one un-pipelined frame: launch -> complete frame resident on rank 0
    fence()
    t1 = time.perf_counter()
    step(0, False)
    fence()
    frame_latency_ms = (time.perf_counter() - t1) * 1e3
    r.synchronize()

    # (after the last use of the recorded frames: rendering another band rebuilds the tile lists they
    # were recorded with)
    # host side of a frame: K frames enqueued back to back with no event and no synchronisation in
    # between -- the queue never fills at this depth, so the loop's wall time is host work only.
    # For the whole frame and for rank 0's share of an 8-rank frame (where it matters: that share is
    # a few tens of microseconds of GPU work).
    host = None
    if world == 1 and a.stage != "bvh":
        host = {"what": "host microseconds per frame, K frames enqueued without waiting: `plain` = "
                        "esc_render_strips (parameter block + one launch per kernel), `recorded` = "
                        "esc_frame_launch (one hipGraphLaunch); through the Python binding",
                "timed_steps_use": "recorded" if a.graph else "plain"}
        share = torch.zeros(multigpu.max_local_rows(H, 8, S) * W * 3, dtype=ch_dtype, device=dev)
        # (--profile-run: whole frames only, so that the trace's average k_frame duration is a frame's)
        for label, fs, stride, out in ((("whole_frame", 0, 1, local[0]),) if a.profile_run else
                                       (("whole_frame", 0, 1, local[0]), ("rank_0_of_8", 0, 8, share))):
            kw = dict(out_f32=None if use_u8 else out, out_u8=out if use_u8 else None, strip_rows=S,
                      shadows=shadows, stage=stage, flags=timed_flags)
            n_rep = min(max(a.steps, 20), 100)  # (well below the depth at which the queue blocks the host)
            with torch.cuda.stream(st):
                rec = r.record_strips(cam, W, H, fs, stride, **kw)
                r.synchronize()
                th = time.perf_counter()
                for _ in range(n_rep):
                    rec.launch()
                t_rec = (time.perf_counter() - th) / n_rep * 1e6
                r.synchronize()
                th = time.perf_counter()
                for _ in range(n_rep):
                    r.render_strips(cam, W, H, fs, stride, **kw)
                t_plain = (time.perf_counter() - th) / n_rep * 1e6
                r.synchronize()
                gpu_us = {}
                for form in ("recorded", "plain"):
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record(st)
                    for _ in range(n_rep):
                        if form == "recorded":
                            rec.launch()
                        else:
                            r.render_strips(cam, W, H, fs, stride, **kw)
                    ev1.record(st)
                    r.synchronize()
                    gpu_us[form] = ev0.elapsed_time(ev1) / n_rep * 1e3
                rec.close()
            host[label] = {"plain_us_per_frame": t_plain, "recorded_us_per_frame": t_rec,
                           "gpu_us_per_frame_back_to_back": gpu_us["plain"],
                           "gpu_us_per_frame_back_to_back_recorded": gpu_us["recorded"]}
        host["host_us_per_frame"] = host["whole_frame"]["recorded_us_per_frame" if a.graph
                                                        else "plain_us_per_frame"]

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
        ppath = prof["paths"].get(a.path) if prof else None
        traffic = ppath.get("hbm_bytes_per_frame") if ppath else None
        algorithm = {
            "bvh": "bounding-volume tree + screen / light bins (ESC_STAGE_BVH, opt-in)",
            "linear": "linear sweep of the primitive table in the reference's index order "
                      "(ESC_RENDER_INDEX_ORDER): every (ray, primitive) pair is decided by the "
                      "reference arithmetic or by a proven conservative filter in front of it -- "
                      "BASELINE's brute-force intersect",
            "culled": ("culled sweep of the primitive table (not brute force): per camera, every "
                       "32x4-pixel tile lists the primitives its primary rays can touch; per one-point "
                       "light, every direction cell of a cube map around it lists the sphere / triangle "
                       "pair records a shadow ray can reach (projected bounds grown by the reference's "
                       "rounding reach, csrc/rt_lists.h); listed primitives go through the proven "
                       "filters and the reference arithmetic; what the lists cannot serve -- "
                       "overflowing tiles / cells, multi-point lights, rays that start outside the "
                       "scene box -- sweeps 3 levels of spatial groups (8 / 64-128 / 512-1,024 "
                       "primitives: DESIGN.md 3.6-3.7); tables under 64 primitives are swept linearly"),
            "lds": "ESC_STAGE_LDS: primitives staged through LDS chunks, the reference arithmetic for every "
                   "(ray, primitive) pair -- no filters, no groups, no lists (the A/B path)",
        }[a.stage if a.stage in ("bvh", "lds") else a.path]
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
                            f"{'1 shadow ray per hit pixel' if shadows else 'no shadow rays'}; "
                            f"{algorithm}",
                "path": a.stage if a.stage in ("bvh", "lds") else a.path,
                "scene": "esc_scene_synthetic (SURVEY.md 8(d)): splitmix64 seeds 0xC2..0xC4",
                "partition": f"{S}-row strips round-robin over {world} rank(s)",
                "gather": ("none (1 GPU)" if world == 1 else
                           f"{'RCCL' if a.backend == 'nccl' else 'gloo'} gather of "
                           f"{'u8' if use_u8 else 'fp32'} RGB strips to rank 0 + k_assemble_strips, "
                           f"on a second stream overlapping the next frame's render; consecutive "
                           f"frames alternate between two render streams"),
                "stage": a.stage,
                "rays_counted": ("in every timed frame (--count-every-frame)" if a.count_every_frame else
                                 "on ONE frame before the timed region, x steps: the library's ray counters "
                                 "are instrumentation the reference does not have (two barriers + atomics "
                                 "per workgroup, 9 % of a c4 frame); the timed frames render the same "
                                 "scene, camera, strips and options with ESC_RENDER_NO_COUNTERS"),
                "rays_per_frame": rays / a.steps,
                "primary_rays_per_frame": primary / a.steps,
                "shadow_rays_per_frame": shadow / a.steps,
                "hit_pixels_per_frame": hits / a.steps,
                "pairs_per_frame_closest_hit": primary / a.steps * (n_tri + n_sph),
                "anyhit_tests_per_frame": anyhit / a.steps,
                "anyhit_tests_per_frame_index_order": anyhit_index_order,
                "shadow_lane_efficiency": (anyhit / lane_tests) if lane_tests else None,
            },
            "kernel": {"name": ("one frame = k_frame (closest hit and shading of a 64x8 tile in one kernel; "
                                "linear sweeps of long lists: k_primary + k_shadow_setup + k_anyhit_segment per "
                                "segment + k_shade_finish per light), HIP events on the launch stream around "
                                "each frame")
                               + ("" if world == 1 else "; N>1: two frames are in flight on two "
                                  "streams, so this duration includes time shared with the other frame"),
                       "avg_ms": kernel_ms,
                       "launches_timed": len(events), "rank": 0, "rows": my_rows},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_source": (prof["source"] if ppath else
                                            "null: profiles/current.json is missing or was "
                                            "measured on other sources (stamp mismatch)"),
                         "traffic_note": "algorithmic = fp32 framebuffer + scene tables (SURVEY.md "
                                         "8(d)); measured traffic = rocprofv3 --pmc FETCH_SIZE / "
                                         "WRITE_SIZE passes, fetch doubled per the gfx950 note; the "
                                         "kernels are VALU-bound, not HBM-bound: see `valu`"},
        }
        if host is not None:
            out["host"] = host
        if kernel_split is not None:
            out["kernel_split"] = kernel_split
        # what bounds the kernels: committed rocprofv3 counters of the shipped sources (stamped)
        if prof:
            valu = {"source": prof["source"] + " (rocprofv3 --pmc, source-stamped; tools/profile.sh)",
                    "definitions": "valu_busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / "
                                   "8 XCDs); wait shares = SQ_WAIT_ANY, SQ_WAIT_INST_ANY / "
                                   "SQ_WAVE_CYCLES; instructions are wave instructions"}
            for pname, pp in prof["paths"].items():
                valu[pname] = {k: counter_view(v) for k, v in pp.get("kernels", {}).items()}
                # per pixel of the frame: wave instructions x 64 lanes / pixels, all kernels of the path
                tot_v = sum(v.get("SQ_INSTS_VALU", 0.0) for v in pp.get("kernels", {}).values())
                tot_s = sum(v.get("SQ_INSTS_SALU", 0.0) for v in pp.get("kernels", {}).values())
                valu[pname]["valu_lane_insts_per_pixel"] = tot_v * 64.0 / (W * H)
                valu[pname]["salu_insts_per_64_pixels"] = tot_s * 64.0 / (W * H)
            out["valu"] = valu
        if linear is not None:
            lin_bytes = alg_bytes
            linear["roofline"] = {"bound": "hbm", "achieved": lin_bytes / (linear["ms_per_step"] * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": lin_bytes / (linear["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": (prof["paths"]["linear"].get("hbm_bytes_per_frame")
                                              if prof and "linear" in prof["paths"] else None)}
            linear["pairs_per_second"] = ((primary / a.steps * (n_tri + n_sph) + anyhit_index_order)
                                          / (linear["ms_per_step"] * 1e-3))
            linear["note"] = ("the same frame, every (ray, primitive) pair in the reference's index "
                              "order through the filters + the reference arithmetic: no groups, no "
                              "re-ordered last light; the brute-force point BASELINE config 5 names")
            out["linear"] = linear
        if pipelined is not None:
            pipelined["value"] = rays / a.steps / (pipelined["ms_per_step"] * 1e-3) / 1e6
            out["pipelined"] = pipelined
        if world == 1 and a.cpu_rows != 0:
            gpu_frame = None
            if not use_u8:
                gpu_frame = local[0][:H * W * 3].cpu().numpy().reshape(H, W, 3)
            cb, same = cpu_baseline(scene, eye, look, W, H, shadows, a.cpu_rows, gpu_frame)
            out["cpu_baseline"] = cb
            # like for like: the CPU baseline sweeps every pair linearly, as `linear` does
            out["gpu_vs_cpu"] = {
                "linear_over_cpu": (linear["value"] / cb["value"]) if linear else None,
                (a.stage if a.stage in ("bvh", "lds") else a.path) + "_over_cpu": out["value"] / cb["value"]}
            out["parity_sample_rows_bit_exact"] = same
        if world == 1 and a.stage != "bvh" and not a.no_accel and not use_u8:
            out["accel"] = accel_leg(esc, r, st, cam, eye, look, W, H, shadows, a.steps, a.warmup,
                                     local[0], alg_bytes, a.count_every_frame)
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
