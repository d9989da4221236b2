#!/bin/bash
# The three tier thresholds, swept on the rotating batches (8 different 37 676-pair samples): rebuilds the library on the GPU box with
# one definition changed at a time and runs the image leg of bench.py.  -> gpurun_out/threshold_sweep.txt
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
one() {   # $1 = label, $2 = EXTRA
  make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 EXTRA="$2" > gpurun_out/sweep_build.log 2>&1 || { echo "$1: build failed"; return 1; }
  a=$(timeout -k 10 200 python bench.py --no-sweep --no-cpu-baseline --rotate-batches --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f %.2f %.4f' % (d['value']/1e6, d['same_batch_every_step']['value']/1e6, d['pi_latency_ms']))")
  b=$(timeout -k 10 200 python bench.py --no-sweep --no-cpu-baseline --rotate-batches --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f %.2f %.4f' % (d['value']/1e6, d['same_batch_every_step']['value']/1e6, d['pi_latency_ms']))")
  echo "$1 | rotating / fixed M images/s, latency ms | run 1: $a | run 2: $b"
}
{
G=${1:-all}
if [ $G = all -o $G = dc ]; then for v in 96 128 160 224 320 1000000; do one "TLC_DC_MIN_POS=$v" "-DTLC_DC_MIN_POS=$v"; done; fi
if [ $G = all -o $G = mh ]; then for v in 64 96 120 150 200 1000000; do one "TLC_MH_MIN_POS=$v" "-DTLC_MH_MIN_POS=$v"; done; fi
if [ $G = all -o $G = tiny ]; then for v in "8 12" "12 16" "14 20" "16 24"; do set -- $v; one "TINY cut n<=$1 m<=$2" "-DTLC_T_NCUT=$1 -DTLC_T_MCUT=$2"; done; fi
} 2>&1 | tee gpurun_out/threshold_sweep_${1:-all}.txt
make -C tlc-gnn_amd/csrc clean > /dev/null; make -C tlc-gnn_amd/csrc -j16 > /dev/null 2>&1
