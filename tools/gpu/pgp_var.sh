#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_chain_batch.py tests/test_gpu_fullsize_oracle.py tests/test_gpu_config5.py tests/test_gpu_concurrency.py -m gpu -x -q > $O/tv.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tv.log
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
for w in 8 4; do
  for spec in "warm 8" "warm 32"; do
    set -- $spec
    echo "== WPG=$w $1 $2 chains, one stream"; BFMMM_PGP_WAVES=$w BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L $1 $2
  done
  echo "== WPG=$w throughput (two streams)"; BFMMM_PGP_WAVES=$w python tools/prof_workload.py --workload warm --chains 8 --steps 300; BFMMM_PGP_WAVES=$w python tools/prof_workload.py --workload warm --chains 32 --steps 200
done
