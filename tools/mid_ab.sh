#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# small outputs: 32 x 32 tiles with eight waves per tile on / off (EG_GEMM_NO_PAIR=1 turns every pair form off)
S="512x512x512 384x384x384 384x512x300 500x500x1000 512x512x2048 448x512x4096 512x512x65536 416x448x256"
echo "== sustained nn"; SPIN_MS=100 python tools/sweep_auto.py nn $S 2>&1 | grep "^nn"
echo "== sustained nn, EG_GEMM_NO_PAIR=1"; SPIN_MS=100 EG_GEMM_NO_PAIR=1 python tools/sweep_auto.py nn $S 2>&1 | grep "^nn"
echo "== sustained tn"; SPIN_MS=100 python tools/sweep_auto.py tn 512x512x512 512x512x65536 2>&1 | grep "^tn"; SPIN_MS=100 EG_GEMM_NO_PAIR=1 python tools/sweep_auto.py tn 512x512x512 512x512x65536 2>&1 | grep "^tn"
