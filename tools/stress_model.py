"""Stress harness for the model path (one parameter update off by 47 % in ~30
full-suite runs).

One process, one context — like the test session — runs a random sequence of small training
scenarios for a wall-clock budget.  Every scenario is executed TWICE on the GPU with two fresh
models (the second one re-uses whatever memory the first one freed), each doing
predict -> loss -> N train steps (eager, captured, replayed), and once on the oracle:

  * the two GPU runs must agree BIT FOR BIT (the backend uses no float atomics: any difference is a
    race, an uninitialised read or a stale captured argument),
  * both must match the oracle at the tolerance of the corresponding test.

Between scenarios the context's scratch blocks are churned (library calls that grow / reuse the
workspace) so captured graphs see their keys change.  Every mismatch is logged with the tensors
involved and the per-step deltas; the summary line ends the log.

    python tools/stress_model.py --seconds 600 --seed 1 --log gpurun_out/stress.jsonl
    EG_POISON=1 / EG_NO_GRAPH=1 / EG_NO_OVERLAP=1 / EG_NO_ROWFUSE=1 python tools/stress_model.py ...
"""
import argparse
import json
import os
os.environ.setdefault("EG_TUNING", "1")   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import exprgrad_amd as eg  # noqa: E402
from exprgrad_amd import dsl, examples, layers, ops  # noqa: E402
from exprgrad_amd import model as egm  # noqa: E402

ACTS = {"relu": layers.relu, "leaky_relu": layers.leaky_relu, "sigmoid": layers.sigmoid, "tanh": layers.tanh}


def mlp(act, dims, rate=0.05):
    net = dsl.input("x")
    for i in range(len(dims) - 1):
        net = layers.dense(net, dims[i], dims[i + 1])
        if i + 2 < len(dims):
            net = ACTS[act](net)
    net = net.target("predict")
    net = layers.mse(net, dsl.input("y")).target("loss")
    return [net.backprop(layers.gradient_descent(rate)).target("train")]


def scenarios():
    out = []
    for act in sorted(ACTS):
        for batch in (64, 37, 300):
            out.append(dict(name=f"mlp-{act}-{batch}", graphs=lambda act=act: mlp(act, (96, 80, 72, 8)), batch=batch,
                            n_in=96, n_out=8, onehot=False, env={"EG_EPILOGUE_MIN_ELEMS": "0"}, prange=0.3, train="train"))
    out.append(dict(name="softmax-2048", graphs=lambda: examples.dense_softmax_net(64, 512, 10), batch=2048, n_in=64,
                    n_out=10, onehot=True, env={}, prange=0.1, train="train"))
    out.append(dict(name="softmax-5000", graphs=lambda: examples.dense_softmax_net(200, 136, 10), batch=5000, n_in=200,
                    n_out=10, onehot=True, env={}, prange=0.1, train="train"))
    out.append(dict(name="xor-1000", graphs=lambda: examples.xor_from_scratch(), batch=1000, n_in=2, n_out=1,
                    onehot=False, env={}, prange=0.5, train="train"))
    out.append(dict(name="xorlayers-4096", graphs=lambda: examples.xor_layers(), batch=4096, n_in=2, n_out=1,
                    onehot=False, env={}, prange=0.5, train="train"))
    out.append(dict(name="mlp-wide-1024", graphs=lambda: mlp("relu", (256, 1024, 16)), batch=1024, n_in=256, n_out=16,
                    onehot=False, env={}, prange=0.1, train="train"))
    return out


def make_data(sc, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((sc["batch"], sc["n_in"]), dtype=np.float32) - 0.5).astype(np.float32)
    if sc["onehot"]:
        y = np.eye(sc["n_out"], dtype=np.float32)[rng.integers(0, sc["n_out"], size=sc["batch"])]
    else:
        y = rng.random((sc["batch"], sc["n_out"]), dtype=np.float32)
    return x, y


def param_values(shapes, seed, prange):
    rng = np.random.default_rng(seed + 7919)
    return {t: ((rng.random(shapes[t], dtype=np.float32) * 2 - 1) * prange).astype(np.float32) for t in sorted(shapes)}


def run_gpu(ctx, sc, seed, steps, sync_every_step):
    for k, v in sc["env"].items():
        os.environ[k] = v
    try:
        m = egm.compile(*sc["graphs"](), gpu=ctx)
        shapes = {t: m._param_shapes[t] for t in m.params.ids()}
        vals = param_values(shapes, seed, sc["prange"])
        for t, v in vals.items():
            m.params[t] = v
        x, y = make_data(sc, seed)
        res = {"predict": m.call("predict", {"x": x}), "loss": m.call("loss", {"x": x, "y": y}), "steps": []}
        for s in range(steps):
            m.apply(sc["train"], {"x": x, "y": y})
            if sync_every_step or s == steps - 1:
                res["steps"].append({t: m.params[t] for t in sorted(shapes)})
        res["plan"] = m.launch_plan(sc["train"])
        m.close()
        return res
    finally:
        for k in sc["env"]:
            os.environ.pop(k, None)


_oracle_cache = {}


def run_oracle(sc, seed, steps, sync_every_step, threads):
    key = (sc["name"], seed, steps, sync_every_step, threads)
    if key in _oracle_cache:
        return _oracle_cache[key]
    import refcases
    from oracle import kd
    ref = kd.Model(refcases.program_text(sc["graphs"]()), threads=threads)
    shapes = {t: list(ref.params[t].shape) for t in ref.params}
    vals = param_values(shapes, seed, sc["prange"])
    for t, v in vals.items():
        ref.params[t][...] = v
    x, y = make_data(sc, seed)
    res = {"predict": ref.call("predict", {"x": x}), "loss": ref.call("loss", {"x": x, "y": y}), "steps": []}
    for s in range(steps):
        ref.apply(sc["train"], {"x": x, "y": y})
        if sync_every_step or s == steps - 1:
            res["steps"].append({t: ref.params[t].copy() for t in sorted(shapes)})
    res["init"] = vals
    _oracle_cache[key] = res
    return res


def rel_err(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    d = np.max(np.abs(want)) if want.size else 0.0
    if not want.size:
        return 0.0
    if not np.all(np.isfinite(got)):
        return float("inf")
    return float(np.max(np.abs(got - want)) / d) if d else float(np.max(np.abs(got)))


def churn(ctx, rng):
    """Library calls between scenarios: split-K contractions of random size (workspace growth /
    reuse), a column sum — what the other test files do between two model tests."""
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return
    M, N = int(rng.integers(2, 200)), int(rng.integers(2, 200))
    K = int(rng.integers(1000, 60000 if kind == 3 else 8000))
    a = ctx.allocTensor((K, M))
    b = ctx.allocTensor((K, N))
    c = ctx.allocTensor((M, N))
    a.write(rng.random((K, M), dtype=np.float32))
    b.write(rng.random((K, N), dtype=np.float32))
    ops.sgemm(ctx, M, N, K, a, M, b, N, c, N, trans_a=True)
    c.read()
    for t in (a, b, c):
        t.buffer.dealloc()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--log", default="gpurun_out/stress.jsonl")
    ap.add_argument("--oracle-threads", type=int, default=4)
    ap.add_argument("--tol", type=float, default=2e-5)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
    log = open(args.log, "a")
    ctx = eg.newGpuContext()
    rng = np.random.default_rng(args.seed)
    scs = scenarios()
    t0 = time.time()
    n_iter = n_bits = n_oracle = n_det = 0
    seen_det = set()
    toggles = {k: os.environ.get(k) for k in ("EG_POISON", "EG_NO_GRAPH", "EG_NO_OVERLAP", "EG_NO_ROWFUSE") if os.environ.get(k)}
    while time.time() - t0 < args.seconds:
        sc = scs[int(rng.integers(0, len(scs)))]
        seed = int(rng.integers(0, 4))          # few data seeds per scenario: the oracle result is cached
        sync = bool(rng.integers(0, 2))
        churn(ctx, rng)
        a = run_gpu(ctx, sc, seed, args.steps, sync)
        churn(ctx, rng)
        b = run_gpu(ctx, sc, seed, args.steps, sync)
        ref = run_oracle(sc, seed, args.steps, sync, args.oracle_threads)
        n_iter += 1
        problems = []
        for key in ("predict", "loss"):
            if not np.array_equal(a[key], b[key], equal_nan=False):
                problems.append({"kind": "gpu-vs-gpu", "what": key, "err": rel_err(a[key], b[key])})
            e = rel_err(a[key], ref[key])
            if not e <= args.tol:
                problems.append({"kind": "gpu-vs-oracle", "what": key, "err": e})
        for s, (pa, pb, pr) in enumerate(zip(a["steps"], b["steps"], ref["steps"])):
            prev = ref["init"] if s == 0 else ref["steps"][s - 1]
            for t in pa:
                if not np.array_equal(pa[t], pb[t]):
                    problems.append({"kind": "gpu-vs-gpu", "what": f"param {t} after read {s}", "err": rel_err(pa[t], pb[t])})
                # compare the UPDATE (what the test compares), relative to its own magnitude
                du_ref = pr[t].astype(np.float64) - prev[t]
                for tag, got in (("A", pa[t]), ("B", pb[t])):
                    du = got.astype(np.float64) - prev[t]
                    scale = np.max(np.abs(du_ref))
                    e = float(np.max(np.abs(du - du_ref)) / scale) if scale > 0 and np.all(np.isfinite(du)) else (
                        0.0 if np.array_equal(du, du_ref) else float("inf"))
                    # the parameters themselves carry ~6e-8 relative rounding: allow it in the difference
                    bound = args.tol + 4e-7 * float(np.max(np.abs(pr[t]))) / scale if scale > 0 else args.tol
                    if not e <= bound:
                        problems.append({"kind": "gpu-vs-oracle", "what": f"run {tag} update of param {t} at read {s}",
                                         "err": e, "bound": bound})
        n_bits += any(p["kind"] == "gpu-vs-gpu" for p in problems)
        # A deviation from the oracle that both runs reproduce bit for bit is not what this harness hunts
        # (races, stale arguments, uninitialised reads): it is either a defect the test suite's f64-shadow
        # rule would catch in any single run, or a relu / max kink — a pre-activation within rounding
        # distance of 0 flips sides with the summation order (softmax-2048, data seed 0: column 352 of the
        # hidden layer, one sample in 2048).  Logged once per (scenario, seed), counted separately.
        det_only = problems and all(p["kind"] == "gpu-vs-oracle" for p in problems)
        if det_only:
            n_det += 1
            if (sc["name"], seed) in seen_det:
                continue
            seen_det.add((sc["name"], seed))
        else:
            n_oracle += any(p["kind"] == "gpu-vs-oracle" for p in problems)
        if problems:
            rec = {"iter": n_iter, "t": round(time.time() - t0, 1), "scenario": sc["name"], "seed": seed, "sync": sync,
                   "problems": problems, "plan": a["plan"], "toggles": toggles}
            log.write(json.dumps(rec) + "\n")
            log.flush()
            print("MISMATCH", json.dumps(rec)[:600], flush=True)
    summary = {"summary": True, "iterations": n_iter, "gpu_vs_gpu_mismatches": n_bits, "gpu_vs_oracle_mismatches": n_oracle,
               "deterministic_deviations_from_oracle": n_det, "deterministic_deviation_cases": sorted(f"{a}/{b}" for a, b in seen_det),
               "seconds": round(time.time() - t0, 1), "seed": args.seed, "steps": args.steps, "toggles": toggles,
               "model_runs": 2 * n_iter}
    log.write(json.dumps(summary) + "\n")
    log.close()
    print(json.dumps(summary))
    return 1 if (n_bits or n_oracle) else 0


if __name__ == "__main__":
    sys.exit(main())
