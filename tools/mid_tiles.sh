#!/bin/bash
# What the contraction planner chooses for the mid sizes and what each candidate is estimated at: tools/mid_tiles.sh
export EG_TUNING=1
for n in 1024 1280 1536 1792 2048 2304; do
  echo "== $n^3"
  EG_DEBUG_TILE=1 SPIN_MS=50 python tools/sweep_auto.py nn ${n}x${n}x${n} 2>&1 | tail -8
done
