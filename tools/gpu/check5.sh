#!/bin/bash
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
bash tools/gpu/check1.sh ${1:-21} 2>&1 | tail -14
echo "== config4"; bash tools/kstat.sh $L config4 1
echo "== config4 again"; bash tools/kstat.sh $L config4 1 | grep chi
