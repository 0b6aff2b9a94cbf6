#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/t4.log 2>&1; echo "pytest rc=$?"; tail -4 $O/t4.log
L=$GRAFT_REPO_ROOT/bayesfmmm_amd/libbfmmm_hip.so
echo "== warm 1"; bash tools/kstat.sh $L warm 1
echo "== warm 8 (one stream)"; BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L warm 8
echo "== warm 32 (one stream)"; BFMMM_BATCH_SPLIT=1 bash tools/kstat.sh $L warm 32
python tools/prof_workload.py --workload warm --chains 8 --steps 300; python tools/prof_workload.py --workload warm --chains 32 --steps 200; python tools/prof_workload.py --workload nu_z --chains 8 --steps 300
