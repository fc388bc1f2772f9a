#!/bin/bash
# Build a variant of the library for A/B experiments: egt_block.hip recompiled with extra flags, the other
# objects reused from egt_amd/lib.  Usage: tools/build_variant.sh <name> [flags...]  ->  egt_amd/lib/var/libegt_<name>.so
# Run with EGT_AMD_LIB=egt_amd/lib/var/libegt_<name>.so (egt_amd/_lib.py).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p egt_amd/lib/var
src=${EGT_VARIANT_SRC:-egt_block.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c egt_amd/csrc/$src -o egt_amd/lib/var/$name.o -I egt_amd/csrc -I include -Wno-unused-result -Wno-pass-failed "$@"
objs=$(ls egt_amd/lib/*.hip.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o egt_amd/lib/var/libegt_$name.so egt_amd/lib/var/$name.o $objs
echo egt_amd/lib/var/libegt_$name.so
