#!/bin/bash
# k_curve_chi per 8-chain batch at M = 6 (46.8 KB of LDS: three workgroups per CU) against M = 5 (40.4 KB: four)
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
for m in 6 5 4; do
  echo "== M=$m warm 8 (one stream)"; KSTAT_ARGS="--M $m" BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L warm 8
  echo "== M=$m warm 1"; KSTAT_ARGS="--M $m" bash tools/kstat.sh $L warm 1 | grep chi
done
