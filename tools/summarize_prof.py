"""Condense a gpurun_out/prof_* directory (rocprofv3 kernel trace + PMC passes written by
tools/profile.sh or tools/prof_bvh.sh) into the small files kept under profiles/:

  kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, as written
  pmc_by_kernel.json   per kernel: mean per launch of every counter + the dispatch's resources
  (with --current)     profiles/current.json: per-FRAME numbers of the shipped kernels, stamped
                       with bench.py's source_stamp() -- bench.py refuses it when the stamp differs

usage: python tools/summarize_prof.py <gpurun_out/prof_X> <profiles/rNN_X> [--current c4]
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
ks = glob.glob(os.path.join(src, "kt", "**", "kt_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
out = {}
calls = {}
for f in glob.glob(os.path.join(src, "pmc_*", "**", "pmc_counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "esc::" not in k:
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

if "--current" in sys.argv:
    config = sys.argv[sys.argv.index("--current") + 1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_stamp", os.path.join(root, "bench.py"))
    # only the stamp function is needed; avoid importing torch: re-implement by reading the source
    src_txt = open(os.path.join(root, "bench.py")).read()
    ns = {"os": os, "ROOT": root}
    start = src_txt.index("def source_stamp():")
    end = src_txt.index("def committed_profile")
    exec(src_txt[start:end], ns)
    stamp = ns["source_stamp"]()

    def frame_kernels(name):  # brute-force frame kernels only (the BVH leg has its own)
        return ("k_primary" in name or "k_shadow_setup" in name or "k_anyhit_segment" in name or
                "k_shade_finish" in name or "k_shade<1" in name or "k_shade<2" in name)

    prim = [k for k in out if "k_primary" in k]
    n_frames = {c: out[prim[0]][c]["launches"] for c in out[prim[0]] if c != "_dispatch"} if prim else {}

    def per_frame(counter, pred):
        tot = 0.0
        for k, v in out.items():
            if not pred(k) or counter not in v:
                continue
            tot += v[counter]["mean_per_launch"] * v[counter]["launches"] / n_frames[counter]
        return tot

    cur = {"source_stamp": stamp, "config": config, "source": os.path.relpath(dst, root)}
    if "FETCH_SIZE" in n_frames and "WRITE_SIZE" in n_frames:
        # rocprofv3 reports KB; FETCH_SIZE counts 64-byte requests of 128-byte lines on gfx950 ->
        # doubled, as the guide's HBM section prescribes
        cur["hbm_bytes_per_frame"] = int((2 * per_frame("FETCH_SIZE", frame_kernels) +
                                          per_frame("WRITE_SIZE", frame_kernels)) * 1024)
        cur["fetch_kb_per_frame"] = per_frame("FETCH_SIZE", frame_kernels)
        cur["write_kb_per_frame"] = per_frame("WRITE_SIZE", frame_kernels)
    if "SQ_INSTS_VALU" in n_frames:
        cur["valu_insts_per_frame"] = {
            "k_primary": per_frame("SQ_INSTS_VALU", lambda k: "k_primary" in k),
            "k_shade": per_frame("SQ_INSTS_VALU", lambda k: frame_kernels(k) and "k_primary" not in k)}
    if "GRBM_GUI_ACTIVE" in n_frames and ks:
        # effective clock of the long kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs
        for r in csv.DictReader(open(ks[0])):
            if "k_primary" in r["Name"]:
                ga = out[prim[0]]["GRBM_GUI_ACTIVE"]["mean_per_launch"]
                # the trace's MINIMUM duration is the undisturbed launch (the bench also runs
                # two-frames-in-flight legs whose launches take longer)
                cur["clock_ghz"] = round(ga / 8.0 / float(r["MinNs"]), 3)
                cur["clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / k_primary's shortest launch"
    json.dump(cur, open(os.path.join(root, "profiles", "current.json"), "w"), indent=1)
    print("profiles/current.json:", cur)
