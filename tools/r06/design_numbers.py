"""Regenerates the ONE table of current numbers in DESIGN.md (between the numbers:begin / numbers:end markers) from the bench
lines collected by tools/collect_profiles.sh + tools/summarize_profile.py:  profiles/<name>_{GINet,sGAT,FoutNet}_benchline.json
    python tools/r06/design_numbers.py r06_v5"""
import json
import os
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
name = sys.argv[1] if len(sys.argv) > 1 else "r06_v5"


def load(net, suffix="benchline"):
    p = os.path.join(root, "profiles", "%s_%s_%s.json" % (name, net, suffix))
    return json.load(open(p)) if os.path.exists(p) else None


def us(v):
    return "—" if v is None else "%.2f" % v


g = load("GINet")
rows = []
rf = g["roofline"]
rows.append(("source hash of the measured tree", rf.get("source_hash", "?")))
rows.append(("**headline** (`value`): GINet SYN64, rebuilt every step, one mini-batch replayed",
             "**%.2f µs per step = %.2f M graphs/s**" % (g["ms_per_step"] * 1e3, g["value"] / 1e6)))
drv = load("GINet", "benchline_driver_args")
if drv:
    rows.append(("… under the driver's arguments (`--steps 20 --warmup 5`)", "%.2f µs = %.2f M graphs/s" % (drv["ms_per_step"] * 1e3, drv["value"] / 1e6)))
rows.append(("dominant kernel `%s`" % rf["kernel"].split(" (")[0], "%.2f µs (HIP events) → %.0f GB/s = **%.1f %%** of 8 TB/s; whole step %.1f %%" % (
    rf["kernel_us"], rf["achieved"], 100 * rf["frac"], 100 * rf.get("whole_step_frac", 0))))
if rf.get("traffic"):
    rows.append(("counter traffic of that launch (FETCH×2 + WRITE)", "%.2f MB = %.2f× the algorithmic 10.10 MB; MFMA utilisation %.1f %%" % (
        rf["traffic"] / 1e6, rf["traffic"] / (157876 * 64), 100 * (rf.get("mfma_util") or 0))))
if rf.get("traffic_cached_topology"):
    rows.append(("… of the launch without a builder (cached topology, NeuralNet's default mode)", "%.2f MB = %.2f×" % (
        rf["traffic_cached_topology"] / 1e6, rf["traffic_cached_topology"] / (157876 * 64))))
db = g.get("distinct_batches") or {}
rows.append(("`distinct_batches` (cycle of 32 different mini-batches)", "%s µs" % us(db.get("us_per_step"))))
ep = g.get("epoch_loop") or {}
rows.append(("`epoch_loop`, NeuralNet default (cached per set), 64 mini-batches per epoch", "%s µs per mini-batch (rebuilt: %s; 1024 per epoch: %s cached / %s rebuilt)" % (
    us(ep.get("us_per_batch")), us((ep.get("rebuilt_topology") or {}).get("us_per_batch")),
    us((ep.get("long_epochs_cached") or {}).get("us_per_batch")), us((ep.get("long_epochs") or {}).get("us_per_batch")))))
nt = g.get("neuralnet_train") or {}
if nt.get("us_per_batch"):
    rows.append(("`neuralnet_train`: `NeuralNet(file, GINet).train(20)` end to end, %d graphs, batch %d" % (nt["graphs"], nt["batch"]),
                 "%.2f µs per mini-batch (%.2f M graphs/s), host bookkeeping included" % (nt["us_per_batch"], nt["graphs_per_s"] / 1e6)))
inf = g.get("inference_loop") or {}
for k, v in inf.items():
    if isinstance(v, dict) and "us_per_batch" in v:
        rows.append(("`inference_loop` %s" % k, "%s µs per mini-batch (%.1f M graphs/s)" % (us(v["us_per_batch"]), v.get("graphs_per_s", 0) / 1e6)))
for net in ("sGAT", "FoutNet"):
    o = load(net)
    if o:
        r = o["roofline"]
        rows.append(("%s SYN64, own bench line" % net, "%.2f µs per step; `%s` %.2f µs = %.1f %%; traffic %s" % (
            o["ms_per_step"] * 1e3, r["kernel"].split(" (")[0], r["kernel_us"], 100 * r["frac"],
            ("%.2f× algorithmic" % (r["traffic"] / (r["alg_bytes_per_graph"] * 64))) if r.get("traffic") else "—") + (
            (" (builder inside the launch; %.2f× without it: cached topology)" % (r["traffic_cached_topology"] / (r["alg_bytes_per_graph"] * 64)))
            if r.get("traffic_cached_topology") else "")))
    elif g.get("other_nets", {}).get(net):
        o = g["other_nets"][net]
        rows.append(("%s SYN64 (`other_nets`)" % net, "%.2f µs per step; kernel %.2f µs = %.1f %%" % (o["us_per_step"], o["kernel_us"], 100 * o["frac"])))
dl = g.get("dropin_loop") or {}
for key in ("GINet", "sGAT", "FoutNet", "GINet_b128"):
    d = dl.get(key)
    if isinstance(d, dict) and "graph_kept_us" in d:
        rows.append(("`dropin_loop` %s: recorded, workspace kept" % key,
                     "model only %.2f µs (%.2f× native %.2f); + MSE + fused Adam %.2f (%.2f×); + torch's default Adam %.1f; rebuilt per call + fused Adam %.2f; eager %.0f" % (
                         d["graph_kept_model_only_us"], d["model_only_over_native"], d["native_distinct_us"], d["graph_kept_fused_adam_us"],
                         d["graph_kept_fused_adam_over_native"], d["graph_kept_us"], d["graph_rebuilt_fused_adam_us"], d["eager_kept_us"])))
cb = g.get("cpu_baseline") or {}
if cb:
    rows.append(("`cpu_baseline` (oracle, `kind: %s`)" % cb.get("kind"), "%.0f graphs/s on %s cores (%s)" % (cb["value"], cb.get("cores"), str(cb.get("sample"))[:80])))
table = "| quantity | measured |\n|---|---|\n" + "\n".join("| %s | %s |" % r for r in rows)
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
a, b = s.index("<!-- numbers:begin"), s.index("<!-- numbers:end -->")
a = s.index("-->", a) + 3
s = s[:a] + "\n" + table + "\n" + s[b:]
open(p, "w").write(s)
print(table)
