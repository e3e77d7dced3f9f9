#!/usr/bin/env python
"""A PREDICTED 1 / 2 / 4 / 8 GPU table for the data-parallel dense step, from ONE GPU — labelled as a model, written so
that the first real N > 1 run (tools/scale_series.py, or the driver's SCALE run) can falsify it.

Measured here (this box, one MI355X): the step without exchange at every shard size the three readings of north_star's
scaling clause need —
    weak      65 536 samples per GPU at every N           (bench.py's default N > 1 line)
    strong-L  524 288 samples over N GPUs                 (BASELINE configs[4] taken literally)
    strong-S  65 536 samples over N GPUs                  ("the dense training step at batch = 65 536 scales >= 6x")
Modelled: the exposed part of the gradient exchange.  eg_model_step_dp reduces the early gradients (W2, b2: 5 130 floats)
on the side lane under the last long contraction and the late piece (W1 with b1 as its last row: 401 920 floats =
1.61 MB) behind it, so one all-reduce of 1.61 MB is exposed per step:
    t_exposed(N) = ALPHA + 2 (N - 1) / N x bytes / BETA            (ring all-reduce over xGMI, N > 1)
with ALPHA = 25 us (launch + 2 (N - 1) ring steps of a latency-bound RCCL call; RCCL on 8 x MI300-class GPUs is usually
quoted at 20 - 40 us for messages of this size) and BETA = 100 GB/s effective per ring direction (xGMI links are
~153 GB/s raw each, MI355X_MICROARCH / task brief).  Both constants are in the JSON; nothing here was measured on more
than one GPU.

    python tools/predict_scaling.py profiles/scaling_prediction_r06.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHA_US, BETA_GBS = 25.0, 100.0
LATE_FLOATS = 784 * 512 + 512
SIZES = [524288, 262144, 131072, 65536, 32768, 16384, 8192]


def step_ms(batch, steps):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "train", "--batch", str(batch),
           "--steps", str(steps), "--no-extra", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
    for text in out.stdout.splitlines():
        if text.startswith("{") and '"metric"' in text:
            d = json.loads(text)
            return d["ms_per_step"], d["roofline"]["frac"]
    raise RuntimeError(out.stdout[-400:] + out.stderr[-400:])


def exposed_ms(n):
    return 0.0 if n == 1 else (ALPHA_US + 2.0 * (n - 1) / n * LATE_FLOATS * 4 / (BETA_GBS * 1e3)) * 1e-3


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "scaling_prediction.json")
    measured = {}
    for b in SIZES:
        ms, frac = step_ms(b, 12 if b >= 262144 else 30)
        measured[b] = {"ms_per_step": ms, "mfma_frac": frac, "samples_per_s": round(b / ms * 1e3, 1)}
        print("measured", b, measured[b], flush=True)
    readings = {"weak (65 536 per GPU)": lambda n: 65536, "strong-L (524 288 global)": lambda n: 524288 // n,
                "strong-S (65 536 global)": lambda n: 65536 // n}
    tables = {}
    md = ["| reading | GPUs | shard | measured shard step (ms) | modelled exposed exchange (ms) | predicted step (ms) | predicted samples / s | predicted speed-up vs 1 GPU |",
          "|---|---|---|---|---|---|---|---|"]
    for name, shard_of in readings.items():
        rows, base = [], None
        for n in (1, 2, 4, 8):
            shard = shard_of(n)
            t = measured[shard]["ms_per_step"] + exposed_ms(n)
            rate = shard * n / t * 1e3
            base = base or rate
            rows.append({"n_gpus": n, "shard": shard, "shard_step_ms": measured[shard]["ms_per_step"], "exposed_exchange_ms": round(exposed_ms(n), 4),
                         "predicted_step_ms": round(t, 4), "predicted_samples_per_s": round(rate, 1), "predicted_speedup": round(rate / base, 3)})
            md.append(f"| {name} | {n} | {shard} | {measured[shard]['ms_per_step']} | {exposed_ms(n):.4f} | {t:.4f} | {rate:.0f} | {rate / base:.2f}x |")
        tables[name] = rows
    out = {"label": "MODEL, not a measurement: one-GPU step times at the shard sizes + a stated all-reduce cost; falsified or confirmed by the "
                    "first tools/scale_series.py run on an 8-GPU node",
           "all_reduce_model": {"formula": "ALPHA + 2 (N - 1) / N * bytes / BETA, one exposed call per step (the late piece)",
                                "alpha_us": ALPHA_US, "beta_gb_per_s": BETA_GBS, "exposed_floats": LATE_FLOATS,
                                "hidden": "the early piece (5 130 floats) runs on the side lane under the last long contraction"},
           "measured_on_one_gpu": {str(k): v for k, v in measured.items()}, "predicted": tables, "markdown": md}
    json.dump(out, open(dst, "w"), indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    main()
