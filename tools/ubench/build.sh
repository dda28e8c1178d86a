#!/bin/bash
# Builds the tools-only microbenchmarks into tools/ubench/bin/ (git-ignored; they travel with gpurun).
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ubench/bin
H=/opt/rocm/bin/hipcc
$H -O2 --offload-arch=gfx950 tools/ubench/library_baseline.cpp -o tools/ubench/bin/library_baseline -lhipfft
$H -O3 --offload-arch=gfx950 -ffp-contract=off -I gpuaudiobench_amd/csrc -I include tools/ubench/mfma_dft.hip -o tools/ubench/bin/mfma_dft
$H -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o tools/ubench/bin/valu_rate

$H -O3 --offload-arch=gfx950 tools/ubench/link_duplex.hip -o tools/ubench/bin/link_duplex
$H -O3 --offload-arch=gfx950 tools/ubench/link_modes.hip -o tools/ubench/bin/link_modes
$H -O3 --offload-arch=gfx950 tools/ubench/dep_chain.hip -o tools/ubench/bin/dep_chain

$H -O2 --offload-arch=gfx950 tools/ubench/prearmed_copy.hip -o tools/ubench/bin/prearmed_copy
$H -O3 --offload-arch=gfx950 -ffp-contract=off -I gpuaudiobench_amd/csrc -I include tools/ubench/far_transform.hip -o tools/ubench/bin/far_transform
$H -O2 --offload-arch=gfx950 tools/ubench/torn_word.hip -o tools/ubench/bin/torn_word
ls -la tools/ubench/bin
