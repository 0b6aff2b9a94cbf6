#!/bin/bash
# k_curve_chi with parts cut out (tools/abl: -DBFMMM_ABLATE build), single chain and the 8-chain batch on one stream
L=$GRAFT_REPO_ROOT/tools/abl/libbfmmm_hip.so
for a in 0 4 8 12 16 20; do
  echo "== BFMMM_ABLATE=$a warm 1"; BFMMM_ABLATE=$a bash tools/kstat.sh $L warm 1 | grep curve_chi
done
for a in 0 4 2 8 16; do
  echo "== BFMMM_ABLATE=$a warm 8"; BFMMM_ABLATE=$a BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L warm 8 | grep curve_chi
done
