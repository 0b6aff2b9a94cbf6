#!/bin/bash
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_tensor.py tests/test_gpu_hd_entry.py tests/test_gpu_shapes.py -m gpu -x -q > $O/thd.log 2>&1; echo "pytest rc=$?"; tail -4 $O/thd.log
bash tools/gpu/hd_prof.sh
