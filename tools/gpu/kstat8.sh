#!/bin/bash
# per-kernel averages of the 8-chain warm batch (one stream), the 32-chain one and the Nu_Z batch
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
for spec in "warm 8" "warm 32" "nu_z 8" "warm 1"; do
  set -- $spec
  echo "== $1 $2 chains, one stream"; BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L $1 $2
done
