#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_fullsize_oracle.py tests/test_gpu_fullsize.py tests/test_gpu_chain_batch.py -m gpu -x -q > $O/t2.log 2>&1; echo "pytest rc=$?"; tail -5 $O/t2.log
python tests/perf/scale_n.py --sizes 4096,16384,65536,262144 > $O/scale_n.json 2> $O/scale_n.err; cat $O/scale_n.json | cut -c1-400
