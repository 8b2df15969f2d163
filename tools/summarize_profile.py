"""Turn rocprofv3 CSV output (gpurun_out/<tag>/, written by tools/collect_profiles.sh) into the tracked summaries
under profiles/.

usage: python tools/summarize_profile.py <name> <gpurun_out/tag dir> [net]
writes profiles/<name>_kernel_stats.csv (the rocprofv3 --stats table, top rows),
       profiles/<name>_summary.md,
       profiles/<name>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE averages, FETCH doubled per MI355X_MICROARCH.md),
       profiles/<name>_sq.json  (per-kernel averages of the SQ / GRBM counters of the extra passes),
       profiles/<name>_benchline.json, <name>_benchline_driver_args.json (the bench lines of the same build).
The two json summaries carry `_meta` = {source_hash, net}: bench.py reports roofline.traffic / mfma_util from them
only when the hash matches the sources it runs from."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

name, src = sys.argv[1], sys.argv[2]
net = sys.argv[3] if len(sys.argv) > 3 else "GINet"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
out = os.path.join(root, "profiles")


def find(sub, suffix):
    hits = sorted(glob.glob(os.path.join(src, sub, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def source_hash():
    import bench
    return bench.source_hash()


meta = {"source_hash": source_hash(), "net": net,
        "command": "tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats / --pmc passes of bench.py --net %s)" % net}
stats = find("stats", "kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, name + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows[:25]:
        w.writerow(r)


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


pmc, cached = {}, {}
fetch, write = find("fetch", "counter_collection.csv"), find("write", "counter_collection.csv")
if fetch and write:
    fe, wr = agg(fetch, "FETCH_SIZE"), agg(write, "WRITE_SIZE")
    for k in fe:
        if "k_" in k[:12]:
            pmc[k] = {"FETCH_SIZE_KB_avg": fe[k], "WRITE_SIZE_KB_avg": wr.get(k),
                      "hbm_bytes_per_launch": (2.0 * fe[k] + (wr.get(k) or 0.0)) * 1024.0,
                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); "
                              "WRITE_SIZE uncalibrated; Infinity-Cache hits are counted"}
    # the launches WITHOUT a co-launched builder (bench.py --topology cached: the default mode of NeuralNet.train), own passes;
    # kept under one key of their own: bench.py's roofline.traffic stays the headline launch's
    cached = {}
    fc, wc = find("fetch_cached", "counter_collection.csv"), find("write_cached", "counter_collection.csv")
    if fc and wc:
        fe_c, wr_c = agg(fc, "FETCH_SIZE"), agg(wc, "WRITE_SIZE")
        for k in fe_c:
            if "k_" in k[:12]:
                cached[k] = {"FETCH_SIZE_KB_avg": fe_c[k], "WRITE_SIZE_KB_avg": wr_c.get(k),
                             "hbm_bytes_per_launch": (2.0 * fe_c[k] + (wr_c.get(k) or 0.0)) * 1024.0}
    extra = {"_cached_topology": cached} if cached else {}
    json.dump(dict(pmc, _meta=meta, **extra), open(os.path.join(out, name + "_pmc.json"), "w"), indent=1)
sq = collections.defaultdict(dict)
for sub in ("sq1", "sq2"):
    path = find(sub, "counter_collection.csv")
    if not path:
        continue
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "k_" in r["Kernel_Name"][:12]:
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        for c, x in v.items():
            sq[k][c] = sum(x) / len(x)
if sq:
    json.dump(dict(sq, _meta=meta), open(os.path.join(out, name + "_sq.json"), "w"), indent=1)
for b in ("benchline.json", "benchline_driver_args.json"):
    if os.path.exists(os.path.join(src, b)) and os.path.getsize(os.path.join(src, b)) > 0:
        shutil.copy(os.path.join(src, b), os.path.join(out, name + "_" + b))
with open(os.path.join(out, name + "_summary.md"), "w") as f:
    f.write("# %s -- rocprofv3 --kernel-trace --stats -- python bench.py --net %s --no-cpu-baseline --epoch-graphs 0 --no-dropin\n\n" % (name, net))
    f.write("kernel sources hash `%s`\n\n" % meta["source_hash"])
    f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    if pmc:
        f.write("\n## HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n")
        f.write("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | corrected bytes/launch |\n|---|---|---|---|\n")
        for k, v in pmc.items():
            f.write("| `%s` | %.1f | %.1f | %.0f |\n" % (k[:60], v["FETCH_SIZE_KB_avg"], v["WRITE_SIZE_KB_avg"] or 0, v["hbm_bytes_per_launch"]))
    if pmc and cached:
        f.write("\n## The same counters, cached topology (bench.py --topology cached: no builder inside the launch)\n\n")
        f.write("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | corrected bytes/launch |\n|---|---|---|---|\n")
        for k, v in cached.items():
            f.write("| `%s` | %.1f | %.1f | %.0f |\n" % (k[:60], v["FETCH_SIZE_KB_avg"], v["WRITE_SIZE_KB_avg"] or 0, v["hbm_bytes_per_launch"]))
    if sq:
        # MI355X: 256 CUs x 4 SIMDs; SQ_WAVE_CYCLES / WAIT / ACTIVE count in the same unit (ratios are unit free);
        # SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over SIMDs, compared with SIMDs x kernel cycles
        # taken from the --stats average duration at the measured 2.28 GHz shader clock
        dur = {r["Name"]: float(r["AverageNs"]) for r in rows}
        f.write("\n## SQ counters per launch (rocprofv3 --pmc, own passes; averages over the launches of a bench run)\n\n")
        f.write("| kernel | MFMA busy cycles | MFMA util (of 1024 SIMDs x kernel cycles) | wave cycles: waiting (waitcnt/barrier) | "
                "issue stall | issuing | LDS bank-conflict / LDS active | VALU insts | LDS insts | SALU insts |\n"
                "|---|---|---|---|---|---|---|---|---|---|\n")
        for k, v in sq.items():
            wc = v.get("SQ_WAVE_CYCLES") or 1.0
            cyc = dur.get(k, 0.0) * 2.28 * 1024.0
            f.write("| `%s` | %.0f | %s | %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.0f | %.0f | %.0f |\n" % (
                k[:60], v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0),
                ("%.2f %%" % (100.0 * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / cyc)) if cyc else "n/a",
                100.0 * v.get("SQ_WAIT_ANY", 0.0) / wc, 100.0 * v.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                100.0 * v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                100.0 * v.get("SQ_LDS_BANK_CONFLICT", 0.0) / (v.get("SQ_LDS_IDX_ACTIVE") or 1.0),
                v.get("SQ_INSTS_VALU", 0.0), v.get("SQ_INSTS_LDS", 0.0), v.get("SQ_INSTS_SALU", 0.0)))
print("wrote profiles/%s_*" % name)
