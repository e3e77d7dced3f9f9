#!/bin/bash
# library's own choice over mid-size and ragged shapes, sustained clocks; pair kernel on / off
S="1000x1000x1000 1000x1024x4096 960x1000x200 1016x1016x1016 900x900x900 512x500x512 1024x1024x1000"
echo "== sustained nn"; SPIN_MS=100 python tools/sweep_auto.py nn $S 1024x1024x1024 2>&1 | grep "^nn"
echo "== sustained nn, EG_GEMM_NO_PAIR=1"; SPIN_MS=100 EG_GEMM_NO_PAIR=1 python tools/sweep_auto.py nn $S 2>&1 | grep "^nn"
echo "== sustained tn"; SPIN_MS=100 python tools/sweep_auto.py tn 1000x1000x1000 2>&1 | grep "^tn"; SPIN_MS=100 EG_GEMM_NO_PAIR=1 python tools/sweep_auto.py tn 1000x1000x1000 2>&1 | grep "^tn"
