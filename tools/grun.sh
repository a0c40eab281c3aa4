#!/bin/bash
# Local wrapper: rebuild the in-tree library (stale .so files travel to the GPU box as they are), then gpurun.
# Usage: bash tools/grun.sh <timeout> '<command>'
cd "$(dirname "$0")/.." && python -c "from exavatar_release_amd import build as b; b.build()" && make -s -C oracle/c && \
  /usr/local/graft/bin/gpurun --timeout $1 -- "$2"
