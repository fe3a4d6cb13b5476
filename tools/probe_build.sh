#!/bin/bash
# Builds variants of the library for kernel experiments, here in the container (hipcc cross-compiles): gpurun_out is not shipped, so they go to
# minimap2-fpga_amd/variants/<name>.so (git-ignored, shipped).   usage: tools/probe_build.sh name "extra hipcc flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/../minimap2-fpga_amd"
mkdir -p variants
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../include -Icsrc -Wall -Wno-unused-result $FLAGS -c csrc/chain_kernel.hip -o variants/$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$NAME.so variants/$NAME.o csrc/chain_epilogue.o csrc/seed_hits.o csrc/host_stage.o csrc/mm2chain_api.o csrc/mm2chain_host.o csrc/mm2chain_seeds.o csrc/mm2chain_dropin.o csrc/mm_chain_dp_host.o csrc/anchor_stream.o -lpthread
  rm -f variants/$NAME.o
  echo "built variants/$NAME.so"
done
