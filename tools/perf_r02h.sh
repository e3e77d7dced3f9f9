#!/bin/bash
# ragged-K tail on the interior loop + remainder rows / columns as contractions of their own
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_conv_grad.py -m gpu -q -x 2>&1 | tail -12 | grep -E "passed|failed|Error"
for v in A=1 EG_GEMM_NO_REMAINDER=1; do
  echo "## $v"
  for mode in nn tn nt tt; do env $v python tools/gemm_shape.py 4100 4100 4100 $mode 5; done
  env $v python tools/gemm_sweep_k.py 4096 4096 nn 4100
  env $v python tools/gemm_sweep_k.py 4100 4100 nn 4100
  env $v python tools/gemm_sweep_k.py 4100 4100 tn 4100
  env $v python tools/gemm_sweep_k.py 8200 8200 nn 8200
  env $v python tools/gemm_sweep_k.py 4128 4100 nn 2052
  env $v python tools/gemm_sweep_k.py 2052 2052 nn 2052
done
