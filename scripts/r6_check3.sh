#!/bin/bash
# round 6, third GPU pass: the whole GPU suite with stderr kept (an abort in the DDP tests was seen once in the second pass)
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s > gpurun_out/r06_check3_pytest_full.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_check3_pytest_full.txt
tail -60 gpurun_out/r06_check3_pytest_full.txt | cut -c1-400
