#!/bin/bash
# usage (on the GPU box): exp/profile_all.sh TAG
# rocprofv3 kernel stats of the default bench, then FETCH_SIZE and WRITE_SIZE in PMC passes of their own;
# summaries land in gpurun_out/prof_TAG/ (copy what is judged into profiles/).
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 30 --warmup 2 --cpu-frames 0 --no-secondary > $out/bench_under_rocprof.json 2> $out/stats.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary > $out/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary > $out/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $out/sq -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary > $out/sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-events --no-secondary > $out/grbm.log 2>&1 || exit 1
python3 exp/mkprofiles.py $tag
