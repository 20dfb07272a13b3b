#!/bin/bash
# A/B of two builds of the library on one box for the training step: scripts/probe/libscgg_before.so against the in-tree build
for v in ${ORDER:-before after before after}; do
  if [ $v = before ]; then export SCGG_LIB_PATH=$PWD/scripts/probe/libscgg_before.so; else unset SCGG_LIB_PATH; fi
  timeout -k 10 200 python scripts/bench_train.py --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],3), round(d['host_issue_ms_per_step'],3), d['loss_last'])" || exit 1
done
