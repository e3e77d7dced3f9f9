"""Debug aid: dense(K -> N) (+ tanh) at batch B, fused and plain, against numpy; garbage in freed memory."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import exprgrad_amd as eg
from exprgrad_amd import dsl, layers, ops
from exprgrad_amd import model as egm

ctx = eg.newGpuContext()
K, N, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
def graphs(act):
    net = layers.dense(dsl.input("x"), K, N)
    if act:
        net = layers.tanh(net)
    return [net.target("predict")]
rng = np.random.default_rng(0)
for trial in range(12):
    junk = torch.full((64 << 20,), float(trial + 7.5), device="cuda"); del junk   # freed memory holds garbage
    for act in (True, False):
        for me in ("0", str(1 << 40)):
            os.environ["EG_EPILOGUE_MIN_ELEMS"] = me
            m = egm.compile(*graphs(act), gpu=ctx)
            vals = {}
            for t in m.params.ids():
                vals[t] = (rng.random(m._param_shapes[t], dtype=np.float32) - 0.5).astype(np.float32)
                m.params[t] = vals[t]
            x = (rng.random((B, K), dtype=np.float32) - 0.5).astype(np.float32)
            got = m.call("predict", {"x": x})
            ws = [v for v in vals.values() if v.ndim == 2][0]; bs = [v for v in vals.values() if v.ndim == 1][0]
            want = x.astype(np.float64) @ ws.astype(np.float64) + bs
            if act:
                want = np.tanh(want)
            err = np.abs(got - want)
            bad = np.argwhere(err > 1e-4)
            if len(bad):
                print(f"trial {trial} act={act} min_elems={me}: {len(bad)} bad, rows {bad[:,0].min()}..{bad[:,0].max()} cols {bad[:,1].min()}..{bad[:,1].max()} max {err.max():.3g}")
                print(m.launch_plan("predict"))
            m.close()
print("done")
# ---- what do the bad values look like?
os.environ["EG_EPILOGUE_MIN_ELEMS"] = "0"
for attempt in range(6):
    m = egm.compile(*graphs(True), gpu=ctx)
    vals = {}
    for t in m.params.ids():
        vals[t] = (rng.random(m._param_shapes[t], dtype=np.float32) - 0.5).astype(np.float32)
        m.params[t] = vals[t]
    x = (rng.random((B, K), dtype=np.float32) - 0.5).astype(np.float32)
    got = m.call("predict", {"x": x}).astype(np.float64)
    ws = [v for v in vals.values() if v.ndim == 2][0].astype(np.float64); bs = [v for v in vals.values() if v.ndim == 1][0].astype(np.float64)
    pre = x.astype(np.float64) @ ws
    want = np.tanh(pre + bs)
    bad = np.argwhere(np.abs(got - want) > 1e-4)
    m.close()
    hist = globals().setdefault("hist", [])
    hist.append((x.astype(np.float64), ws, bs))
    if not len(bad):
        continue
    full = np.arctanh(np.clip(got, -0.999999, 0.999999))
    for age, (xo, wo, bo) in enumerate(reversed(hist[:-1]), 1):
        for name, cand in (("old x, old W, old b", xo @ wo + bo), ("new x, old W, new b", x @ wo + bs), ("old x, new W, new b", xo @ ws + bs),
                           ("new x, old W, old b", x @ wo + bo), ("new x, new W, old b", pre + bo)):
            d = np.abs(full - cand)[bad[:, 0], bad[:, 1]]
            print(f"    age {age} {name}: {int((d < 1e-3).sum())} of {len(bad)} bad elements explained")
    r, c = bad[len(bad) // 2]
    r0, c0 = (r // 32) * 32, (c // 32) * 32
    blk = np.s_[r0:r0 + 32, c0:c0 + 32]
    acc = np.arctanh(np.clip(got[blk], -0.999999, 0.999999)) - bs[c0:c0 + 32]
    print(f"attempt {attempt}: {len(bad)} bad; block rows {r0}.. cols {c0}..; want pre[0,:4]={pre[blk][0,:4]}, got acc[0,:4]={acc[0,:4]}")
    # candidate explanations
    for name, cand in (("2x", 2 * pre[blk]), ("0", 0 * pre[blk])):
        print("   ", name, float(np.max(np.abs(acc - cand))))
    best = None
    for rr in range(0, B - 31, 32):
        for cc in range(0, N - 31, 32):
            d = float(np.max(np.abs(acc - pre[rr:rr + 32, cc:cc + 32])))
            if best is None or d < best[0]:
                best = (d, rr, cc)
    print("    closest other block of the true product:", best)
    for rr in range(0, B - 31, 32):
        for cc in range(0, N - 31, 32):
            d = float(np.max(np.abs(acc - pre[blk] - pre[rr:rr + 32, cc:cc + 32])))
            if d < 1e-3:
                print("    = own block + block", rr, cc, d)
    if attempt >= 2: break
