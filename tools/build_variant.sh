#!/bin/bash
# build_variant.sh <name> <extra hipcc flags...>  ->  tools/_build/libpnpx_<name>.so  (A/B builds for tools/, PNPX_LIB=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/pnpx_var_XXXX)
mkdir -p $tmp/tfpnp_amd $tmp/include
cp -r $root/tfpnp_amd/csrc $tmp/tfpnp_amd/csrc
cp $root/include/pnpx.h $tmp/include/
rm -f $tmp/tfpnp_amd/csrc/*.o
make -C $tmp/tfpnp_amd/csrc -j16 EXTRA="$*" tuning >/dev/null
mkdir -p $root/tools/_build
cp $tmp/tools/_build/libpnpx_tune.so $root/tools/_build/libpnpx_$name.so
rm -rf $tmp
echo built tools/_build/libpnpx_$name.so with "$*"
