#!/bin/bash
# Wave-cycle breakdown (parked / issue-stalled / issuing) per kernel, one pass of 8 SQ counters.
# usage: tools/profile_sq.sh <tag>
set -u
tag=${1:-sq}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --warmup 2 --cpu-cycles 0 --no-solve --no-fast --steps 3 ${BENCH_EXTRA:-}"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- $B > $out/bench_sq.json 2> $out/sq.err || { echo "sq pass failed"; tail -5 $out/sq.err; exit 1; }
echo "sq $tag done"
