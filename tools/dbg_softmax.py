"""Debug aid: one scenario of tools/stress_model.py, update of every parameter against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import exprgrad_amd as eg
import stress_model as sm

name, seed = sys.argv[1], int(sys.argv[2])
sc = [s for s in sm.scenarios() if s["name"] == name][0]
ctx = eg.newGpuContext()
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a = sm.run_gpu(ctx, sc, seed, steps, True)
ref = sm.run_oracle(sc, seed, steps, True, 4)
print(a["plan"])
for s in range(steps):
    prev = ref["init"] if s == 0 else ref["steps"][s - 1]
    for t in a["steps"][s]:
        du = a["steps"][s][t].astype(np.float64) - prev[t]
        dr = ref["steps"][s][t].astype(np.float64) - prev[t]
        scale = np.max(np.abs(dr))
        err = np.abs(du - dr) / scale
        idx = np.unravel_index(np.argmax(err), err.shape)
        print(f"step {s} param {t} shape {dr.shape}: max err {err.max():.3e} at {idx}, count>1e-3: {(err > 1e-3).sum()}, |p|/|du| {np.max(np.abs(prev[t]))/scale:.0f}")
