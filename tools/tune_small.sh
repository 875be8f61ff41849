#!/bin/bash
# wall-clock sweep of conv_hs launch overrides at small batches (tuning library); usage: tune_small.sh B [keys...]
export PNPX_LIB=tools/_build/libpnpx_tune.so
B=$1; shift
KEYS=${@:-"32_16 32_32 64_32 32_64 64_64 32_128 64_128 32_256"}
run() { env $1 python tools/ab_wall.py "chains=0" $B 256 2>&1 | grep forward | awk '{print $3}' | sort -n | head -1; }
echo "B=$B baseline $(run "A=1") $(run "A=1")"
for key in $KEYS; do
  for cfg in 1,4 2,4 4,4 1,8 2,8; do
    echo "PNPX_HS_$key=$cfg $(run "PNPX_HS_$key=$cfg")"
  done
done
