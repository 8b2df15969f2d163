"""Turn rocprofv3 CSV output (gpurun_out/...) into the tracked summaries under profiles/.

usage: python tools/summarize_profile.py <tag> <kernel_stats.csv> [<fetch counter_collection.csv> <write counter_collection.csv>]
writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --stats table, top rows),
       profiles/<tag>_summary.md, profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE averages)."""
import collections
import csv
import json
import os
import sys

tag, stats = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows[:25]:
        w.writerow(r)
pmc = {}
if len(sys.argv) >= 5:
    def agg(path, counter):
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return {k: sum(v) / len(v) for k, v in d.items()}
    fe, wr = agg(sys.argv[3], "FETCH_SIZE"), agg(sys.argv[4], "WRITE_SIZE")
    for k in fe:
        if "k_" in k[:12]:
            pmc[k] = {"FETCH_SIZE_KB_avg": fe[k], "WRITE_SIZE_KB_avg": wr.get(k),
                      "hbm_bytes_per_launch": (2.0 * fe[k] + (wr.get(k) or 0.0)) * 1024.0,
                      "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); "
                              "WRITE_SIZE uncalibrated; Infinity-Cache hits are counted"}
    json.dump(pmc, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
with open(os.path.join(out, tag + "_summary.md"), "w") as f:
    f.write("# %s -- rocprofv3 --kernel-trace --stats\n\n" % tag)
    f.write("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|\n")
    for r in rows[:16]:
        f.write("| `%s` | %s | %.2f | %.2f | %.2f | %s |\n" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                           float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    if pmc:
        f.write("\n## HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)\n\n")
        f.write("| kernel | FETCH_SIZE KB | WRITE_SIZE KB | corrected bytes/launch |\n|---|---|---|---|\n")
        for k, v in pmc.items():
            f.write("| `%s` | %.1f | %.1f | %.0f |\n" % (k[:60], v["FETCH_SIZE_KB_avg"], v["WRITE_SIZE_KB_avg"] or 0, v["hbm_bytes_per_launch"]))
print("wrote profiles/%s_*" % tag)
