#!/bin/bash
# Build the gfx950 library locally (hipcc cross-compiles), then run a command on an MI355X box.
set -e
cd "$(dirname "$0")/.."
python -m virtex_amd.build > /dev/null
T=${GPU_TIMEOUT:-900}
exec /usr/local/graft/bin/gpurun --timeout $T -- "$@"
