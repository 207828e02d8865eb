#!/bin/bash
for b in "$@"; do
  timeout -k 10 120 python bench.py --steps 5 --warmup 2 --cpu-frames 0 --batch $b 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($b, d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
