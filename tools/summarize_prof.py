"""Condense a gpurun_out/prof_* directory (rocprofv3 kernel trace + PMC passes written by
tools/profile.sh or tools/prof_bvh.sh) into the small files kept under profiles/:

  kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, as written
  pmc_by_kernel.json   per kernel: mean per launch of every counter + the dispatch's resources
  (with --current)     profiles/current.json: per-FRAME numbers of the shipped kernels, stamped
                       with bench.py's source_stamp() -- bench.py refuses it when the stamp differs

usage: python tools/summarize_prof.py <gpurun_out/prof_X> <profiles/rNN_X> [--current c4]
With --current the source holds one sub-directory per path (culled/, linear/: tools/profile.sh) and
profiles/current.json gets, per path, the HBM bytes per frame and every SQ counter per kernel class
(k_primary, k_shade = all shading kernels, k_frame) summed per frame.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def condense(src, dst):
    """one profile directory (kt/ + pmc_*/) -> kernel_stats.csv + pmc_by_kernel.json under dst"""
    os.makedirs(dst, exist_ok=True)
    ks = glob.glob(os.path.join(src, "kt", "**", "kt_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
    out = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "pmc_counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        rows = [r for r in csv.DictReader(open(f)) if "esc::" in r["Kernel_Name"]]
        # bench.py also launches the frame kernels on a few sample rows (parity spot-check) and on rank
        # shares: only launches with the kernel's LARGEST grid are whole frames
        biggest = collections.defaultdict(int)
        for r in rows:
            biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r.get("Grid_Size", 0) or 0))
        for r in rows:
            k = r["Kernel_Name"]
            if int(r.get("Grid_Size", 0) or 0) != biggest[k]:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {x: r[x] for x in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                          "Grid_Size", "Workgroup_Size") if x in r}
        for k, v in agg.items():
            o = out.setdefault(k, {"_dispatch": meta[k]})
            for c, xs in v.items():
                o[c] = {"launches": len(xs), "mean_per_launch": sum(xs) / len(xs)}
    json.dump(out, open(os.path.join(dst, "pmc_by_kernel.json"), "w"), indent=1)
    for k, v in out.items():
        print(k[:60], {c: round(x["mean_per_launch"]) for c, x in v.items() if c != "_dispatch"})
    return out, (ks[0] if ks else None)


def kernel_class(name):
    """frame kernels of the brute-force paths by role (the BVH leg's k_shade<3,..> is not one)"""
    if "k_prepare" in name or "k_bin" in name:
        return "k_prepare"
    if "k_frame" in name:
        return "k_frame"
    if "k_primary" in name:
        return "k_primary"
    if ("k_shadow_setup" in name or "k_anyhit_segment" in name or "k_shade_finish" in name or
            "k_shade<1" in name or "k_shade<2" in name):
        return "k_shade"
    return None


def per_frame(out, path):
    """per-FRAME sums of every counter per kernel class of the path's frame: culled = k_frame alone
    (the k_primary / k_shade launches of that run are bench.py's two-kernel split and are kept out),
    linear = k_primary + the queue-form shading kernels; frames = whole-frame launches of the anchor"""
    want = ("k_frame",) if path == "culled" else ("k_primary", "k_shade")
    out = {k: v for k, v in out.items() if kernel_class(k) in want}
    anchor = [k for k in out if kernel_class(k) == want[0]]
    if not anchor:
        return {}, {}
    frames = {c: sum(out[k][c]["launches"] for k in anchor if c in out[k])
              for c in out[anchor[0]] if c != "_dispatch"}
    cls = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, v in out.items():
        kc = kernel_class(k)
        if kc is None or kc == "k_prepare":
            continue
        for c, x in v.items():
            if c == "_dispatch" or not frames.get(c):
                continue
            cls[kc][c] += x["mean_per_launch"] * x["launches"] / frames[c]
    return {k: dict(v) for k, v in cls.items()}, frames


if "--current" not in sys.argv:
    condense(src, dst)
else:
    config = sys.argv[sys.argv.index("--current") + 1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # bench.py's stamp function without importing torch
    src_txt = open(os.path.join(root, "bench.py")).read()
    ns = {"os": os, "ROOT": root}
    start = src_txt.index("def source_stamp():")
    end = src_txt.index("def committed_profile")
    exec(src_txt[start:end], ns)
    cur = {"source_stamp": ns["source_stamp"](), "config": config,
           "source": os.path.relpath(dst, root), "paths": {}}
    for path in ("culled", "linear"):
        if not os.path.isdir(os.path.join(src, path)):
            continue
        out, _ = condense(os.path.join(src, path), os.path.join(dst, path))
        cls, frames = per_frame(out, path)
        ent = {"kernels": cls}
        if frames.get("FETCH_SIZE") and frames.get("WRITE_SIZE"):
            # rocprofv3 reports KB; FETCH_SIZE counts 64-byte requests of 128-byte lines on gfx950 ->
            # doubled, as the guide's HBM section prescribes
            f = sum(v.get("FETCH_SIZE", 0.0) for v in cls.values())
            w = sum(v.get("WRITE_SIZE", 0.0) for v in cls.values())
            ent.update({"hbm_bytes_per_frame": int((2 * f + w) * 1024), "fetch_kb_per_frame": f,
                        "write_kb_per_frame": w})
        cur["paths"][path] = ent
    json.dump(cur, open(os.path.join(root, "profiles", "current.json"), "w"), indent=1)
    print("profiles/current.json:", json.dumps(cur)[:600])
