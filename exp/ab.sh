#!/bin/bash
# A/B experimental builds of libsgmx.so: usage exp/ab.sh name1 name2 ... (exp/libsgmx_<name>.so; "base" = in-tree)
for n in "$@"; do
  if [ "$n" = base ]; then unset SGMX_LIB; else export SGMX_LIB=$PWD/exp/libsgmx_$n.so; fi
  echo "== $n"
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
