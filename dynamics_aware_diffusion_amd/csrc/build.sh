#!/bin/bash
# Build libdad_hip.so (the C-ABI of include/dad.h) for gfx950, in-tree.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../libdad_hip.so"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wall \
      -o "$out" "$here/dad_lib.hip"
echo "built $out"
