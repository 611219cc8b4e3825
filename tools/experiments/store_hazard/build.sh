#!/bin/bash
# builds the standalone experiment (here; the binary travels to the GPU box with the snapshot)
cd "$(dirname "${BASH_SOURCE[0]}")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 store_hazard.hip -o store_hazard
