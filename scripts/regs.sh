#!/bin/bash
# Register / LDS / scratch use per kernel of one source file: scripts/regs.sh icp.hip [kernel name filter] [extra flags]
SRC=${1:-icp.hip}; F=${2:-k_}; shift 2
cd "$(dirname "$0")/../mandala_mapping_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math "$@" -Rpass-analysis=kernel-resource-usage -c $SRC -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - - | grep "$F"
