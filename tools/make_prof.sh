#!/bin/bash
# A copy of the engine built with -DZPQ_LZ_PROFILE under tools/_prof (git-ignored, travels with gpurun): what tools/lzprof2.py loads
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf $R/tools/_prof
mkdir -p $R/tools/_prof
cp -r $R/include $R/tools/_prof/include
mkdir -p $R/tools/_prof/zpaqfranz_amd
cp $R/zpaqfranz_amd/*.py $R/tools/_prof/zpaqfranz_amd/
cp -r $R/zpaqfranz_amd/csrc $R/zpaqfranz_amd/shim $R/tools/_prof/zpaqfranz_amd/
cd $R/tools/_prof
ZPQ_EXTRA_FLAGS=-DZPQ_LZ_PROFILE python -c "import sys; sys.path.insert(0, '.'); from zpaqfranz_amd import build; build.build(force=True)"
