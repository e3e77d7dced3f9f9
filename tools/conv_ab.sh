#!/bin/bash
# configs[3]'s convolution (CASES=forward | grad_filter | grad_image, default forward) as kernel times from a rocprofv3
# trace, 400 launches back to back: tools/conv_ab.sh [ENV=VALUE ...]   (GPU box).  EG_LIB_PATH=<other build> for an A/B.
bash tools/trace_cmd.sh CASES=${CASES:-forward} "$@" -- python tools/conv_shape.py ${SHAPE:-1 256 256 64 64 3 3} 400 | grep -E "==|conv2|slab|operands"
