#!/bin/bash
# k_curve_chi with parts cut out / moved (tools/abl: -DBFMMM_ABLATE build), single chain
L=$GRAFT_REPO_ROOT/tools/abl/libbfmmm_hip.so
for a in ${ABL_LIST:-0 32 4 0 32}; do
  echo "== BFMMM_ABLATE=$a warm 1"; BFMMM_ABLATE=$a bash tools/kstat.sh $L warm 1 | grep curve_chi
done
