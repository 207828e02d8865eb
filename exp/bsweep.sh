#!/bin/bash
# batch-size sweep: exp/bsweep.sh B1 B2 ...   (frames/s and per-stage ms per step)
for b in "$@"; do
  echo "== B=$b"
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-frames 0 --batch $b 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
