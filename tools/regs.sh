#!/bin/bash
# usage: tools/regs.sh <file stem under r-tucker_amd/csrc> [grep filter]  -- register / spill report of a gfx950 compile
cd /root/repo
mkdir -p /tmp/regs
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Iinclude -Ir-tucker_amd/csrc -Wall -Wno-unused-function \
  -c r-tucker_amd/csrc/$1.hip -o /tmp/regs/$1.o -save-temps=obj 2>&1 | grep -E "error|warning" | head
f=/tmp/regs/$1-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|sgpr_count)|^\s+\.name:" $f | paste - - - - - | sed 's/private_segment_fixed_size/scratch/' | grep -E "${2:-.}"
