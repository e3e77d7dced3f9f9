#!/bin/bash
export EG_TUNING=1   # measurement aids (class `tuning` of csrc/switches.cpp) are honoured only with it
# eg_dgemm: every tile config (EG_DGEMM_TILE=<config>[,<splits>]) and the library's own choice, per size
export PYTHONPATH=.
SIZES=${SIZES:-"512 1024 1536 2048 3072 4096"}
for cfg in auto 0 1 2 "2,2" "2,4" "1,2"; do
  echo "== EG_DGEMM_TILE=$cfg"
  if [ "$cfg" = auto ]; then python tools/dgemm_bench.py $SIZES 2>&1 | grep dgemm
  else EG_DGEMM_TILE=$cfg python tools/dgemm_bench.py $SIZES 2>&1 | grep dgemm; fi
done
