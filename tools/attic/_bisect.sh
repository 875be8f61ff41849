#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in _t_1f51520 _t_fc2e8c5 _t_b5271bc; do
  f=0
  for i in $(seq 1 14); do
    (cd $t && python -m pytest tests/test_gpu_edge.py -q -m gpu -k "two_contexts" 2>&1 | tail -1 | grep -q failed) && f=$((f+1))
  done
  echo "$t: $f failures of 14"
done
