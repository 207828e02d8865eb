#!/bin/bash
# usage: exp/kt_ab.sh script.py "args" name1 name2 ...  -- kernel traces (last 11 sgmx launches) of a script per experimental build
script=$1; sargs=$2; shift 2
for n in "$@"; do
  if [ "$n" = base ]; then unset SGMX_LIB; else export SGMX_LIB=$GRAFT_REPO_ROOT/exp/libsgmx_$n.so; fi
  echo "== $n"
  bash exp/ktrace_cmd.sh kt_ab_$n 11 $script $sargs || exit 1
done
