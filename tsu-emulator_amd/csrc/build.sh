#!/bin/bash
# Build libtsu_hip.so for gfx950 in-tree (the .so travels to the GPU box with the repo snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../tsu/_lib"
mkdir -p "$OUT" "$HERE/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
objs=()
pids=()
for src in "$HERE"/*.hip; do
    obj="$HERE/_obj/$(basename "${src%.hip}").o"
    objs+=("$obj")
    if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/tsu_common.h" -nt "$obj" ] || [ "$HERE/ising2d.h" -nt "$obj" ] || [ "$HERE/dense.h" -nt "$obj" ] || [ "$HERE/dense_dev.h" -nt "$obj" ] || [ "$HERE/../../include/tsu_hip.h" -nt "$obj" ]; then
        $HIPCC $FLAGS ${TSU_EXTRA_FLAGS:-} -c "$src" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libtsu_hip.so.tmp$$" "${objs[@]}"
mv "$OUT/libtsu_hip.so.tmp$$" "$OUT/libtsu_hip.so"
echo "built $OUT/libtsu_hip.so"
