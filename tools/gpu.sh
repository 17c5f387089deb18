#!/bin/bash
# build everything from the sources as they are NOW, then run a command on the GPU box: the box runs the binaries that
# are in the tree, and a library built before the last edit is a silent A/B of the wrong thing.
#   tools/gpu.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
python -c "from audiality2_amd import build; build.build_all()" || exit 1
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
