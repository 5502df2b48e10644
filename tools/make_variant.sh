#!/bin/bash
# A copy of the engine with patches applied and / or extra compiler flags, built under tools/_variants/<name> (git-ignored, travels
# with gpurun): what tools/run_variant.py puts in front of the tree's own package, so that ONE GPU call can time several
# candidates against the tree.   usage: tools/make_variant.sh <name> [-D...|-f...]... [patch file]...
# e.g.  tools/make_variant.sh emit tools/proto/lz77_vector_emitter.patch
#       tools/make_variant.sh emit_prof -DZPQ_LZ_PROFILE tools/proto/lz77_vector_emitter.patch
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
[ -n "$N" ] || { echo "usage: $0 <name> [flags] [patches]"; exit 2; }
V=$R/tools/_variants/$N
rm -rf "$V"
mkdir -p "$V/zpaqfranz_amd"
cp -r "$R/include" "$V/include"
cp "$R"/zpaqfranz_amd/*.py "$V/zpaqfranz_amd/"
cp -r "$R/zpaqfranz_amd/csrc" "$R/zpaqfranz_amd/shim" "$V/zpaqfranz_amd/"
FLAGS=""
for a in "$@"; do
  case "$a" in
    -*) FLAGS="$FLAGS $a" ;;
    *) (cd "$V" && patch -p1 -s < "$(cd "$R" && realpath "$a")") ;;
  esac
done
cd "$V"
ZPQ_EXTRA_FLAGS="$FLAGS" python -c "import sys; sys.path.insert(0, '.'); from zpaqfranz_amd import build; build.build(force=True)"
rm -rf "$V/zpaqfranz_amd/build"
