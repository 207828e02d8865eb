#!/bin/bash
# A/B over an environment switch: usage exp/ab_env.sh VAR v1 v2 ...   (one short bench per value; stops at the first failure)
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-frames 0 ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
