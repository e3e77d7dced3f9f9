"""Four calls of the filter gradient at BASELINE configs[3] (debugging aid: EG_GRADF_TRACE=1 prints the per-wave cycle stamps of
kernels/conv2_gradf_halo.hip — prologue, every segment, the fold).  GPU box: EG_GRADF_TRACE=1 python tools/gf_once.py"""
import sys; sys.path.insert(0,'/root/repo')
import torch, exprgrad_amd as eg
from exprgrad_amd import ops
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = eg.newGpuContext(0, stream=stream.cuda_stream)
N,H,W,C,F=1,256,256,64,64
img = torch.rand((N,H,W,C), device="cuda"); gout = torch.rand((N,H-2,W-2,F), device="cuda")-0.5
gflt = torch.empty((F,3,3,C), device="cuda")
for _ in range(4):
    ops.conv2_nhwc_grad_filter(ctx,N,H,W,C,F,3,3,img,gout,gflt)
torch.cuda.synchronize()
