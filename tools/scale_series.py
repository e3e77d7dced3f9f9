#!/usr/bin/env python
"""north_star's table: throughput of the data-parallel dense step (BASELINE configs[4]) at 1 / 2 / 4 / 8 GPUs as absolute
numbers and as a fraction of the MFMA roofline, with the reference CPU path timed on the host cores of the same box
beside it (core count stated).

    python tools/scale_series.py                 # every GPU count <= the GPUs visible, weak and strong
    python tools/scale_series.py --gpus 1 2 --scaling weak --steps 20
    tools/scale_series.sh                        # the same through bash (what INTEGRATION.md quotes)

Each point is one `python bench.py --gpus N --workload train [--scaling strong]` run (bench.py starts its N ranks
itself); the JSON lines are kept in gpurun_out/scale_series/ and the table is printed as markdown and written to
gpurun_out/scale_series/table.md.  weak = 65 536 samples per GPU (N = 8 is the config's 524 288 global batch);
strong = the 524 288 global batch divided over N GPUs.  Efficiency = value(N) / (N x value(1)) for weak scaling,
speed-up = value(N) / value(1) for strong scaling — computed here from the per-N values only, like the driver does."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def visible_gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


def run_point(n, scaling, steps, warmup, out_dir, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--workload", "train", "--steps", str(steps),
           "--warmup", str(warmup), "--scaling", scaling, "--no-extra"] + extra
    if n > 1:
        cmd.append("--no-cpu-baseline")
    log = os.path.join(out_dir, f"{scaling}_n{n}.log")
    with open(log, "w") as f:
        rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=ROOT)
    line = None
    for text in open(log):
        text = text.strip()
        if text.startswith("{") and '"metric"' in text:
            line = json.loads(text)
    if rc != 0 or line is None:
        return {"error": f"rc {rc}, see {log}"}
    json.dump(line, open(os.path.join(out_dir, f"{scaling}_n{n}.json"), "w"), indent=1)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--scaling", nargs="*", default=["weak", "strong"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scale_series"))
    ap.add_argument("bench_args", nargs="*", help="passed on to bench.py (e.g. --torch-dp)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    have = visible_gpus()
    counts = [n for n in args.gpus if n <= max(have, 1)]
    skipped = [n for n in args.gpus if n > max(have, 1)]
    rows, md = [], []
    for scaling in args.scaling:
        base = None
        md.append(f"\n### {scaling} scaling — dense 784-512-10 train step, "
                  + ("65 536 samples per GPU" if scaling == "weak" else "524 288 samples over all GPUs") + "\n")
        md.append("| GPUs | global batch | ms / step | samples / s | TFLOP/s (all GPUs) | fraction of MFMA f32 peak (per GPU) | vs 1 GPU | "
                  "scaling efficiency | all-reduce calls / step | reference CPU path, same box |")
        md.append("|---|---|---|---|---|---|---|---|---|---|")
        for n in counts:
            line = run_point(n, scaling, args.steps, args.warmup, args.out, args.bench_args)
            if "error" in line:
                md.append(f"| {n} | — | {line['error']} | | | | | | | |")
                continue
            if n == 1:
                base = line
            cfg, roof = line.get("config", {}), line.get("roofline", {})
            speed = line["value"] / base["value"] if base else None
            eff = (speed / n if scaling == "weak" else speed / n) if speed else None
            cpu = line.get("cpu_baseline") or {}
            cpu_text = (f"{cpu.get('value')} {cpu.get('unit')} on {cpu.get('cores')} cores ({cpu.get('kind')})" if cpu.get("value") else "—")
            tf = roof.get("achieved", 0) * n
            md.append(f"| {n} | {cfg.get('global_batch')} | {line['ms_per_step']} | {line['value']:.0f} | {tf:.1f} | {roof.get('frac')} | "
                      f"{speed:.2f}x | {eff:.3f} | {(line.get('exchange') or {}).get('pieces_split', '—')} | {cpu_text} |" if speed else
                      f"| {n} | {cfg.get('global_batch')} | {line['ms_per_step']} | {line['value']:.0f} | {tf:.1f} | {roof.get('frac')} | — | — | — | {cpu_text} |")
            rows.append({"scaling": scaling, "n_gpus": n, "ms_per_step": line["ms_per_step"], "value": line["value"],
                         "unit": line["unit"], "roofline_frac": roof.get("frac"), "speedup_vs_n1": speed, "efficiency": eff})
    if skipped:
        md.append(f"\nNot run: {skipped} GPUs (this box shows {have}).")
    text = "\n".join(md)
    print(text)
    open(os.path.join(args.out, "table.md"), "w").write(text + "\n")
    json.dump(rows, open(os.path.join(args.out, "table.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
