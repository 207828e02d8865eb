#!/bin/bash
# per-launch kernel trace for experimental builds: usage [EXTRA="bench args"] exp/kt_ab.sh name1 name2 ... ("base" = in-tree)
for n in "$@"; do
  if [ "$n" = base ]; then unset SGMX_LIB; else export SGMX_LIB=$GRAFT_REPO_ROOT/exp/libsgmx_$n.so; fi
  bash exp/ktrace.sh kt_$n || exit 1
done
