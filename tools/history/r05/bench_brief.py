"""prints the figures of a bench line that the round's targets are stated in"""
import json
import sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.3f M graphs/s   %.2f us per step   kernel %.2f us  frac %.4f" % (d["value"] / 1e6, d["ms_per_step"] * 1e3, r["kernel_us"], r["frac"]))
for k, v in r["kernels"].items():
    print("   %-78s %6.2f us" % (k[:78], v["avg_us"]))
db = d.get("distinct_batches", {})
print("distinct_batches: same %.2f  distinct %.2f" % (db.get("us_per_step_same_batch", 0), db.get("us_per_step", 0)))
e = d.get("epoch_loop", {})
print("epoch_loop: rebuilt %.2f  cached %.2f  long rebuilt %.2f  long cached %.2f" % (
    e.get("us_per_batch", 0), e.get("cached_topology", {}).get("us_per_batch", 0), e.get("long_epochs", {}).get("us_per_batch", 0),
    e.get("long_epochs_cached", {}).get("us_per_batch", 0)))
for n, v in (d.get("other_nets") or {}).items():
    print("%s: %s" % (n, {k: (round(x, 3) if isinstance(x, float) else x) for k, x in v.items() if k in ("us_per_step", "kernel_us", "frac", "error")}))
