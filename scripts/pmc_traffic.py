"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py`.

    python scripts/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/r01_pmc_traffic.json

Units and corrections as MI355X_MICROARCH.md "HBM" prescribes: rocprofv3 reports both counters in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores.  Both kernels priced here read and write with float4 accesses.
"""
import collections
import csv
import glob
import json
import sys


by_kernel = {"FETCH_SIZE": collections.defaultdict(list), "WRITE_SIZE": collections.defaultdict(list)}


def short(n):
    return n.replace("void ", "").replace("suave::", "").replace("(anonymous namespace)::", "").split("(")[0]


def per_class(path, counter):
    acc = collections.defaultdict(list)
    f = sorted(glob.glob(path + "/**/*_counter_collection.csv", recursive=True))[-1]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        cls = ("gemm" if "gemm" in n else "likelihood_kl" if ("k_loss" in n or "k_wide_softmax" in n) else
               "sample_heads" if ("k_sample_heads" in n or "k_wide_sample" in n) else None)
        if cls:
            acc[cls].append(float(r["Counter_Value"]))
        if "suave" in n:
            by_kernel[counter][short(n)].append(float(r["Counter_Value"]))
    return acc


fetch = per_class(sys.argv[1], "FETCH_SIZE")
write = per_class(sys.argv[2], "WRITE_SIZE")
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 5 --warmup 2 "
                 "--no-cpu-baseline --no-profile`; KiB -> bytes, FETCH_SIZE x2 (gfx950 wide-read correction)"}
for cls in sorted(set(fetch) | set(write)):
    rd = 2.0 * 1024.0 * sum(fetch[cls]) / max(len(fetch[cls]), 1)
    wr = 1024.0 * sum(write[cls]) / max(len(write[cls]), 1)
    out[cls] = {"launches_sampled": len(fetch[cls]), "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
# every library kernel: mean HBM bytes per launch (same corrections) and launches sampled
out["kernels"] = {}
for k in sorted(set(by_kernel["FETCH_SIZE"]) | set(by_kernel["WRITE_SIZE"])):
    f, w = by_kernel["FETCH_SIZE"].get(k, []), by_kernel["WRITE_SIZE"].get(k, [])
    out["kernels"][k] = {"launches": max(len(f), len(w)), "read_MB": round(2.0 * 1024.0 * sum(f) / max(len(f), 1) / 1e6, 3),
                         "write_MB": round(1024.0 * sum(w) / max(len(w), 1) / 1e6, 3)}
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if steps:  # whole training step: bytes of every sampled launch / steps profiled (warm-up included in both)
    out["per_step"] = {cls: {"read_MB": round(2.0 * 1024.0 * sum(fetch[cls]) / steps / 1e6, 1),
                             "write_MB": round(1024.0 * sum(write[cls]) / steps / 1e6, 1)} for cls in sorted(set(fetch) | set(write))}
    out["per_step"]["all_library_kernels"] = {
        "read_MB": round(2.0 * 1024.0 * sum(sum(v) for v in by_kernel["FETCH_SIZE"].values()) / steps / 1e6, 1),
        "write_MB": round(1024.0 * sum(sum(v) for v in by_kernel["WRITE_SIZE"].values()) / steps / 1e6, 1)}
print(json.dumps(out, indent=1))
