#!/bin/bash
# Register / scratch / LDS use of every kernel in one source file (compile only, no GPU needed):
#   tools/kernel_resources.sh k_conv_accel.hip [extra flags]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/gpuaudiobench_amd/csrc/$1"; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I"$ROOT/include" -I"$ROOT/gpuaudiobench_amd/csrc" \
  -x hip -c "$SRC" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|LDS Size|Occupancy" | sed 's/.*remark: [^ ]* //' | sed "s@ \[-Rpass-analysis=kernel-resource-usage\]@@" | paste - - - - -
