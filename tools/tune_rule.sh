#!/bin/bash
export PNPX_LIB=tools/_build/libpnpx_tune.so
run() { env $1 python tools/ab_wall.py "chains=0" $2 256 2>&1 | grep forward | awk '{print $3}' | sort -n | head -1; }
for B in 6 9 10 12 14 18 24 30 38 44; do
  echo "B=$B rule0 $(run PNPX_HS_RULE=0 $B) rule2 $(run PNPX_HS_RULE=2 $B) rule6 $(run PNPX_HS_RULE=6 $B) rule4 $(run PNPX_HS_RULE=4 $B) rule3 $(run PNPX_HS_RULE=3 $B)"
done
