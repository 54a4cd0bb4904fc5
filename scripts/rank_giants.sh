#!/bin/bash
# rank shares of the C3 box at N = 2 and 4 (first rank each) with more of the rank's large halos through the multi-workgroup pipeline
for n in 2 4; do
  for ng in default 262144 131072 65536; do
    if [ $ng = default ]; then unset SOAP_ENGINE_NB_GIANT; else export SOAP_ENGINE_NB_GIANT=$ng; fi
    echo -n "N=$n NB_GIANT=$ng: "; RANKS=0 python scripts/rank_share.py $n 2>&1 | grep "^rank" | sed "s/particles.*calls\/pass 1//" | cut -c1-330
  done
done
