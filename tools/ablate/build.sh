#!/bin/bash
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -I../../include -DRTK_ABLATE_STAMPS -I. -I../../r-tucker_amd/csrc rtk_score_ablate.hip -o librtk_ablate.so
echo built tools/ablate/librtk_ablate.so
