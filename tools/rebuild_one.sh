#!/bin/bash
# Recompile the given csrc units (names without .hip) and relink librtucker_hip.so (the others from obj/).
set -e
cd "$(dirname "$0")/../r-tucker_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function"
for f in "$@"; do hipcc $FLAGS -c $f.hip -o obj/$f.o; done
SRCS=$(grep '^SRCS=' build.sh | cut -d'"' -f2)
objs=""; for f in $SRCS; do objs="$objs obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/librtucker_hip.so.new $objs
mv ../lib/librtucker_hip.so.new ../lib/librtucker_hip.so
echo "relinked"
