#!/bin/bash
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
for w in 4 2; do
echo "== single chain, packed kernel forced, WPG=$w"; BFMMM_PGP_WAVES=$w BFMMM_PG_PACK=1 bash tools/kstat.sh $L warm 1
done
