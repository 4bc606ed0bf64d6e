#!/bin/bash
# Read-request size breakdown at the L2's memory side (32/64/128-byte requests), one pass.
# usage: tools/profile_rd.sh <tag>
set -u
tag=${1:-rd}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --warmup 2 --cpu-cycles 0 --no-solve --steps 3"
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rd -- $B > $out/bench_rd.json 2> $out/rd.err || { echo "rd pass failed"; tail -5 $out/rd.err; exit 1; }
echo "rd $tag done"
