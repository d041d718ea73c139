#!/bin/bash
cd $GRAFT_REPO_ROOT; out=gpurun_out/r3_run10; rm -rf $out; mkdir -p $out
timeout 1500 python -m pytest tests/test_headline_gpu.py -m gpu -q -s -k "assp48_full_size or fp16_gradients" > $out/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $out/summary.txt
cat $out/summary.txt; grep -v "^$" $out/pytest_new.log | tail -25
