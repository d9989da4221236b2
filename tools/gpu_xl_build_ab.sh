#!/bin/bash
# builds of the lane kernels with different wavefronts per workgroup, each A/B'd against xl_cut=0 in one process
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "$@"; do
  (cd tlc-gnn_amd/csrc && touch extract_lane.hip pd_tiny.hip && make -j8 EXTRA="$cfg" > /dev/null 2>&1) || { echo "build failed: $cfg"; continue; }
  echo "== $cfg"
  python tools/ab_option.py xl_cut 0 24 30 2>&1 | grep xl_cut
done
(cd tlc-gnn_amd/csrc && touch extract_lane.hip pd_tiny.hip && make -j8 > /dev/null 2>&1)
