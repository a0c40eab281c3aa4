#!/bin/bash
# Build libexa_raster from the csrc/ + include/ of a git revision into exavatar_release_amd/_variants/<name>.so
# (same flags as exavatar_release_amd/build.py), for A/B runs with EXA_RASTER_LIB on the GPU box.
# Usage: bash tools/build_variant.sh <git-rev> <name> [extra hipcc flags...]
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/exavatar_release_amd/csrc $T/include $ROOT/exavatar_release_amd/_variants
if [ "$REV" = WORK ]; then      # the working tree (e.g. with -D flags that select an experimental code path)
  cp $ROOT/exavatar_release_amd/csrc/* $T/exavatar_release_amd/csrc/; cp $ROOT/include/exa_raster.h $T/include/
else
  for f in $(git -C $ROOT ls-tree --name-only $REV exavatar_release_amd/csrc/); do git -C $ROOT show $REV:$f > $T/$f; done
  git -C $ROOT show $REV:include/exa_raster.h > $T/include/exa_raster.h
fi
OBJS=""
for src in preprocess_fwd binning render_fwd render_bwd compose preprocess_bwd ssim api; do
  [ -f $T/exavatar_release_amd/csrc/$src.hip ] || continue
  X=""; [ $src = preprocess_fwd ] && X="-ffp-contract=off"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -w $X "$@" -c $T/exavatar_release_amd/csrc/$src.hip -o $T/$src.o
  OBJS="$OBJS $T/$src.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/exavatar_release_amd/_variants/$NAME.so $OBJS
rm -rf $T
echo $ROOT/exavatar_release_amd/_variants/$NAME.so
