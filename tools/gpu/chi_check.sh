#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_gpu_shapes.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_multivariate.py tests/test_gpu_covariates.py tests/test_gpu_chain_batch.py tests/test_gpu_exact_instances.py -m gpu -x -q > $O/tc.log 2>&1; echo "pytest rc=$?"; tail -4 $O/tc.log
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
echo "== warm 1"; bash tools/kstat.sh $L warm 1
echo "== warm 8 (one stream)"; BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L warm 8
echo "== config4"; bash tools/kstat.sh $L config4 1
echo "== nu_z 8 (one stream)"; BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L nu_z 8
