#!/bin/bash
# Selected GPU tests with their printed numbers:  bash scripts/run_tests.sh <out-name> <pytest args...>
cd $GRAFT_REPO_ROOT; out=gpurun_out/$1; shift; mkdir -p $out
timeout 2400 python -m pytest "$@" -m gpu -x -q -s > $out/pytest.log 2>&1; echo "rc=$?" > $out/summary.txt
cat $out/summary.txt; grep -v "^$" $out/pytest.log | tail -60
