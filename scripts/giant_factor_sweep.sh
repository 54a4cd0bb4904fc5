#!/bin/bash
# how eager should the giants' mode be?  slowest rank of the C3 box at N = 1, 2, 4, 8 for a few margins
cd "$(dirname "$0")/.."
for f in ${FACTORS:-0.4 0.7 1.0}; do
  for n in ${NS:-1 2 4 8}; do
    echo "factor $f: $(SOAP_ENGINE_GIANT_FACTOR=$f timeout -k 10 400 python scripts/rank_share.py $n c3 2>&1 | grep '^N=')"
  done
done
