"""Reads a rocprofv3 output directory (kernel trace + memory copy trace, csv) of tools/r05/epoch_boundary_probe.py and prints
the device timeline around the epoch boundaries: every record between the last k_update of one epoch and the second step launch
of the next, with start offsets and durations (us), and the mean boundary gap."""
import csv
import glob
import sys


def load(pattern, kind):
    rows = []
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Name") or r.get("Direction") or kind
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, name))
    return rows


def main():
    d = sys.argv[1]
    rec = load(d + "/**/*kernel_trace.csv", "K") + load(d + "/**/*memory_copy_trace.csv", "C")
    rec.sort()
    short = lambda n: n.split("(")[0][:70]
    # boundaries: a record that is neither a step launch nor k_update, after at least 20 step launches in a row
    is_step = lambda n: "co_topo" in n
    is_upd = lambda n: n.startswith("k_update")
    runs, i = [], 0
    streak = 0
    gaps = []
    for j, (s, e, k, n) in enumerate(rec):
        if is_step(n) or is_upd(n):
            streak += 1
            continue
        if streak >= 40:
            # walk back to the last update, forward to the first step launch after the foreign records
            a = j - 1
            b = j
            while b < len(rec) and not is_step(rec[b][3]):
                b += 1
            if b + 2 < len(rec):
                runs.append((a, b))
        streak = 0
    print("%d epoch boundaries found" % len(runs))
    for (a, b) in runs[-3:]:
        t0 = rec[a][1]
        print("---- boundary: last update ends at 0")
        for (s, e, k, n) in rec[a - 2:b + 3]:
            print("  %s start %8.2f  dur %7.2f  %s" % (k, (s - t0) / 1e3, (e - s) / 1e3, short(n)))
    # steady-state period of a mini-batch (step start to step start) against the boundary's
    for (a, b) in runs:
        gaps.append((rec[b][0] - rec[a][1]) / 1e3)
    if gaps:
        print("gap last-update-end -> first step start: mean %.2f us  (min %.2f  max %.2f)" % (sum(gaps) / len(gaps), min(gaps), max(gaps)))
    steps = [r for r in rec if is_step(r[3])]
    per = sorted((steps[i + 1][0] - steps[i][0]) / 1e3 for i in range(len(steps) - 1))
    if per:
        print("step-to-step period: median %.2f us  p10 %.2f  p90 %.2f" % (per[len(per) // 2], per[len(per) // 10], per[9 * len(per) // 10]))


if __name__ == "__main__":
    main()
