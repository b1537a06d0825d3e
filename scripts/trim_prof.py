"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into profiles/<name>.csv (library kernels in
full, everything else summed into one 'other (torch plumbing: table synthesis, randperm)' line)."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
out, other = [], [0, 0]
for r in rows:
    if "suave" in r["Name"]:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        out.append([name.split("(")[0], r["Calls"], r["TotalDurationNs"], "%.1f" % float(r["AverageNs"]),
                    "%.2f" % (100.0 * int(r["TotalDurationNs"]) / tot), r["MinNs"], r["MaxNs"]])
    else:
        other[0] += int(r["Calls"]); other[1] += int(r["TotalDurationNs"])
out.append(["other (torch plumbing: synthetic table, randperm, fills)", other[0], other[1], "%.1f" % (other[1] / max(other[0], 1)),
            "%.2f" % (100.0 * other[1] / tot), "", ""])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    w.writerows(out)
print(open(dst).read())
