#!/bin/bash
# Fixed cost of the halo convolution kernels: kernel time (rocprofv3 trace) of configs[3]'s spatial shape at several
# channel counts (time = fixed + chunks x per-chunk) and with two images (two items per block: what the roll-over hides).
# tools/conv_intercept.sh            (GPU box)
for shape in "1 256 256 16 64" "1 256 256 32 64" "1 256 256 64 64" "1 256 256 128 64" "2 256 256 64 64" "1 256 256 64 128" "1 256 256 64 32"; do
  bash tools/trace_cmd.sh -- python tools/conv_shape.py $shape 3 3 40 | grep -E "==|conv2|slab|operands"
done
