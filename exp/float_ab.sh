#!/bin/bash
# A/B of the float-volume path and the materialised census path: usage exp/float_ab.sh name1 name2 ... ("base" = in-tree)
for n in "$@"; do
  if [ "$n" = base ]; then unset SGMX_LIB; else export SGMX_LIB=$GRAFT_REPO_ROOT/exp/libsgmx_$n.so; fi
  echo "== $n"
  timeout -k 10 120 python exp/float_time.py || exit 1
  timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-frames 0 --no-secondary --p2 300 --batch 8 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('materialised census (P2=300), 8 pairs:', d['value'], 'frames/s', d['roofline']['per_direction_ms'])" || exit 1
done
