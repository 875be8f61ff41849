#!/bin/bash
# Timing-only ablations of conv3x3_wino.hip's pipeline stage: builds libpnpx_abl<k>.so with -DWINO_ABL=<k> (see the source) next to the
# production library and prints the per-layer profile of each.  Run on the GPU box: bash tools/wino_ablate.sh "1 2 4 8 16 31"
cd $GRAFT_REPO_ROOT/tfpnp_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -w -Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=$(ls *.o | grep -v "conv3x3_wino.o\|tune")
for k in ${1:-1 2 4 8 16}; do
  /opt/rocm/bin/hipcc $FL -DWINO_ABL=$k -c conv3x3_wino.hip -o /tmp/wino_abl$k.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/wino_abl$k.o -o /tmp/libpnpx_abl$k.so
  echo "== WINO_ABL=$k"
  (cd $GRAFT_REPO_ROOT && PNPX_LIB=/tmp/libpnpx_abl$k.so python tools/profile_layers.py 48 256 0 2>&1 | tail -42 | head -38 | cut -c1-60 | grep "^ 6 \|^14 \|^21 \|^34 ")
done
