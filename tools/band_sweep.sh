#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# fit step at batch 4096 (fashion_mnist network), kernel times of the band convolutions for several band sizes
for px in 512 1024 2048 4096; do
  echo "== EG_CONV_BAND_PIXELS=$px"
  EG_CONV_BAND_PIXELS=$px FIT_BATCH=4096 bash tools/fit_profile.sh 2>&1 | grep -E "conv_band|colsum" | head -8
done
