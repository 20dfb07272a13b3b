#!/bin/bash
# A/B of two builds of the library on one box: scripts/probe/libscgg_before.so against the in-tree build, alternating
for v in before after before after; do
  if [ $v = before ]; then export SCGG_LIB_PATH=$PWD/scripts/probe/libscgg_before.so; else unset SCGG_LIB_PATH; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), d['config']['gathered_labels_checksum'])" || exit 1
done
