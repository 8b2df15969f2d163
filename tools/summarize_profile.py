"""Turn rocprofv3 CSV output (gpurun_out/...) into the tracked summaries under profiles/.

usage: python tools/summarize_profile.py <tag> <kernel_stats.csv> [<fetch counter_collection.csv> <write counter_collection.csv>
                                          [<SQ counter_collection.csv> ...]]
writes profiles/<tag>_kernel_stats.csv (the rocprofv3 --stats table, top rows),
       profiles/<tag>_summary.md, profiles/<tag>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE averages),
       profiles/<tag>_sq.json (per-kernel averages of the SQ / GRBM counters of the extra passes)."""
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
sq = collections.defaultdict(dict)
for path in sys.argv[5:]:
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if "k_" in r["Kernel_Name"][:12]:
            d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        for c, x in v.items():
            sq[k][c] = sum(x) / len(x)
if sq:
    json.dump(sq, open(os.path.join(out, tag + "_sq.json"), "w"), indent=1)
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
print("wrote profiles/%s_*" % tag)
