"""Condense a gpurun_out/prof_* directory (rocprofv3 kernel trace + PMC passes written by
tools/profile.sh or tools/prof_bvh.sh) into the small files kept under profiles/."""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ks = glob.glob(os.path.join(src, "kt", "**", "kt_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(dst, "kernel_stats.csv"))
out = {}
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
