#!/bin/bash
# Variant of the library with egt_block_bwd6.hip recompiled with extra flags (see tools/build_variant.sh).
# Usage: tools/build_variant6.sh <name> [flags...] -> egt_amd/lib/var/libegt_<name>.so ; run with EGT_AMD_LIB=...
set -e
cd "$(dirname "$0")/.."
EGT_VARIANT_SRC=egt_block_bwd6.hip exec tools/build_variant.sh "$@"
