#!/usr/bin/env python
"""Condense one tools/profile.sh output directory into the small files kept under profiles/.

  python tools/summarize_profile.py gpurun_out/prof_<tag> profiles/<name> [--workload KEY --steps K]

Writes <name>/kernel_stats.csv (rocprofv3 --kernel-trace --stats, backend kernels only, names
shortened) and <name>/pmc_summary.txt (per-kernel per-launch means of every counter that was
collected, one `--pmc` pass per group).  With --workload it also records the HBM traffic of the
workload in profiles/traffic.json, which bench.py reports as `roofline.traffic`:
    traffic = 2 x FETCH_SIZE + WRITE_SIZE   (KiB -> bytes)
FETCH_SIZE is doubled as MI355X_MICROARCH.md's HBM section prescribes for gfx950 (wide coalesced
reads are tallied at 64 B per 128-B request); WRITE_SIZE is taken as reported.  For a
single-kernel workload the traffic is per launch of that kernel; for a whole-step workload it is
the sum over the backend kernels of one step (total over the profiled run / launches of the
step's first kernel).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:140]


def ours(name):
    return (name.startswith("eg") or "eg::" in name or "eg_" in name or
            any(k in name for k in ("colsum", "rowsum", "conv2_halo", "conv2_gradf", "grad_image_operands", "copy_segments", "dgemm_",
                                    "map_kernel", "map_grad_kernel", "bias_add_kernel", "axpy_kernel", "fill_kernel", "sum_partial",
                                    "sum_final", "slab_sum", "zero_ranges", "gemm_streamk")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--kernel", default=None, help="substring of the dominant kernel (single-kernel workloads)")
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    os.makedirs(args.dst, exist_ok=True)

    stats = glob.glob(os.path.join(args.src, "trace", "**", "*kernel_stats.csv"), recursive=True) or \
        glob.glob(os.path.join(args.src, "**", "*kernel_stats.csv"), recursive=True)
    # Per-dispatch records: the mean over ALL launches of a profiled bench run includes the clock-ramp
    # spin-up launches bench.py issues before its warmup (up to 13 % slower), so it sits above the timed
    # ms_per_step.  SteadyAverageNs is the mean over the second half of a kernel's launches in time order
    # (the timed region is the end of the run): that is the figure that must agree with the bench line.
    steady = {}
    traces = glob.glob(os.path.join(args.src, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if traces:
        per_kernel = collections.defaultdict(list)
        for r in csv.DictReader(open(traces[0])):
            per_kernel[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for n, v in per_kernel.items():
            v.sort()
            tail = [d for _, d in v[len(v) // 2:]]
            steady[n] = (sum(tail) / len(tail), len(tail))
    calls = {}
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        with open(os.path.join(args.dst, "kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "SteadyAverageNs", "SteadyCalls"])
            for r in rows:
                n = short(r["Name"])
                if not ours(n):
                    continue
                calls[n] = int(r["Calls"])
                st = steady.get(n, ("", ""))
                w.writerow([n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"],
                            f"{st[0]:.1f}" if st[0] != "" else "", st[1]])

    per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> values
    for path in sorted(glob.glob(os.path.join(args.src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(path)):
            n = short(r["Kernel_Name"])
            if ours(n):
                per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["rocprofv3 PMC passes (one counter group per run; per-launch means). FETCH_SIZE / WRITE_SIZE in KiB as",
             "reported; HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section).",
             args.note, ""]
    totals = collections.Counter()
    for n in sorted(per, key=lambda k: -sum(per[k].get("FETCH_SIZE", [0]))):
        lines.append(n)
        for c in sorted(per[n]):
            v = per[n][c]
            lines.append(f"    {c:32s} n={len(v):5d} mean={sum(v) / len(v):.6g}")
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                totals[c] += sum(v)
    with open(os.path.join(args.dst, "pmc_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")

    if args.workload:
        tpath = os.path.join(os.path.dirname(os.path.abspath(args.dst)), "traffic.json")
        table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        if args.kernel:
            ks = [n for n in per if args.kernel in n and "FETCH_SIZE" in per[n]]
            k = max(ks, key=lambda n: sum(per[n]["FETCH_SIZE"]))
            fetch = sum(per[k]["FETCH_SIZE"]) / len(per[k]["FETCH_SIZE"])
            write = sum(per[k]["WRITE_SIZE"]) / len(per[k]["WRITE_SIZE"])
            entry = {"kernel": k, "launches": len(per[k]["FETCH_SIZE"])}
        else:
            # whole step: launches of the kernel that runs exactly once per step
            once = min((len(v["FETCH_SIZE"]) for v in per.values() if "FETCH_SIZE" in v), default=1)
            fetch = totals["FETCH_SIZE"] / once
            write = totals["WRITE_SIZE"] / once
            entry = {"kernel": "all backend kernels of one step", "launches": once}
        fp_path = os.path.join(args.src, "source_fingerprint.txt")
        if os.path.exists(fp_path):
            entry["source_fingerprint"] = open(fp_path).read().strip()
        entry.update({"fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
                      "traffic_bytes": int((2 * fetch + write) * 1024),
                      "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB", "source": os.path.relpath(args.dst)})
        table[args.workload] = entry
        json.dump(table, open(tpath, "w"), indent=1, sort_keys=True)
        print(args.workload, entry)


if __name__ == "__main__":
    sys.exit(main())
