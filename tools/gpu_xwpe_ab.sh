#!/bin/bash
# extraction kernel capped at 96 VGPRs (five wavefronts per SIMD) with a grid that uses the fifth slot, against the default build
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "" "-DTLC_X_WPE=5"; do
    rm -f tlc-gnn_amd/csrc/build/extract.o; make -C tlc-gnn_amd/csrc -j16 EXTRA="$v" > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $v"; exit 1; }
    echo "EXTRA='$v'"; timeout -k 10 300 python tools/ab_option.py x_grid 0 5120 30 2>&1 | grep x_grid
  done
done | tee gpurun_out/xwpe_ab.txt
rm -f tlc-gnn_amd/csrc/build/extract.o; make -C tlc-gnn_amd/csrc -j16 > /dev/null 2>&1
