#!/bin/bash
# Variant build of the MFMA inner op for A/B runs and measurement builds:
#   bash tools/build_attn_variant.sh <name> [source.hip] [hipcc flags...]   ->  egt_amd/lib/var/libegt_<name>.so
# (every other object comes from the regular build; run with EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_<name>.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=egt_amd/csrc/egt_attn_mfma.hip
if [ -n "${1:-}" ] && [ -f "$1" ]; then SRC=$1; shift; fi
mkdir -p egt_amd/lib/var
TMP=egt_amd/csrc/_variant_$NAME.hip
cp "$SRC" $TMP
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $TMP -o egt_amd/lib/var/attn_$NAME.o -I egt_amd/csrc -I include -Wno-unused-result -w "$@"
rm -f $TMP
OBJS=$(ls egt_amd/lib/*.hip.o | grep -v egt_attn_mfma)
hipcc --offload-arch=gfx950 -shared -fPIC -o egt_amd/lib/var/libegt_$NAME.so $OBJS egt_amd/lib/var/attn_$NAME.o -ldl
echo egt_amd/lib/var/libegt_$NAME.so
