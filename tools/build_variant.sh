#!/bin/bash
# Build a variant of the library for A/B experiments: ONE translation unit recompiled (extra flags and / or another source text),
# the other objects reused from egt_amd/lib.
#   tools/build_variant.sh <name> [hipcc flags...]        ->  egt_amd/lib/var/libegt_<name>.so
#   EGT_VARIANT_SRC=egt_attn_mfma.hip   which translation unit (default egt_block.hip)
#   EGT_VARIANT_FILE=/path/to/old.hip    compile this text in its place (e.g. a saved earlier version)
# Run with EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_<name>.so (egt_amd/_lib.py); tools/ab.sh compares variants on the GPU box.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p egt_amd/lib/var
src=${EGT_VARIANT_SRC:-egt_block.hip}
in=egt_amd/csrc/$src
if [ -n "${EGT_VARIANT_FILE:-}" ]; then in=egt_amd/csrc/_variant_$name.hip; cp "$EGT_VARIANT_FILE" $in; fi
extra=$(python -c "
import sys; sys.path.insert(0, '.')
from egt_amd import build
print(' '.join(build.EXTRA_FLAGS.get('$src', [])))")
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $in -o egt_amd/lib/var/$name.o -I egt_amd/csrc -I include -Wno-unused-result -Wno-pass-failed $extra "$@"
if [ -n "${EGT_VARIANT_FILE:-}" ]; then rm -f $in; fi
objs=$(ls egt_amd/lib/*.hip.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o egt_amd/lib/var/libegt_$name.so egt_amd/lib/var/$name.o $objs -ldl
echo egt_amd/lib/var/libegt_$name.so
