import argparse, sys, json
sys.path.insert(0, '/root/repo')
import bench, torch
import exprgrad_amd as eg
from exprgrad_amd import ops
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
import torch.distributed as dist
clock = None
try:
    clock = bench.DeviceClock(ctx, 0) if "clock" in sys.argv else None
except Exception as e:
    print("clock", e)
timer = bench.Timer(torch, dist, 1, stream, None, clock)
env = {"torch": torch, "ops": ops, "ctx": ctx, "timer": timer}
args = argparse.Namespace(steps=20, warmup=3)
out = bench.run_float64(args, env)
print(json.dumps(out["sizes"]))
print(out["conv2_benchmark"]["ms_per_call"])
