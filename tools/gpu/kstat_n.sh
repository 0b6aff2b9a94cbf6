#!/bin/bash
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
for n in 16384 65536; do
  echo "== n=$n"; KSTAT_ARGS="--n $n" bash tools/kstat.sh $L warm 1 100
done
