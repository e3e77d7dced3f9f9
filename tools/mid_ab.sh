#!/bin/bash
S="256x256x256 384x384x384 512x512x512 640x640x640 768x768x768 1024x1024x256 1024x1024x512 512x512x2048"
echo "== sustained nn auto"; SPIN_MS=100 python tools/sweep_auto.py nn $S 1000x1024x4096 2>&1 | grep "^nn"
echo "== sustained nn unsliced"; SPIN_MS=100 EG_GEMM_FORCE_SPLITS=1 python tools/sweep_auto.py nn $S 2>&1 | grep "^nn"
echo "== sustained nn unsliced 64"; SPIN_MS=100 EG_GEMM_FORCE_SPLITS=1 EG_GEMM_FORCE_TILE=64,64 python tools/sweep_auto.py nn $S 2>&1 | grep "^nn"
