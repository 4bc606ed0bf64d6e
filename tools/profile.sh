#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics and (in separate passes) the HBM
# traffic counters for the benchmark's 513^3 V-cycle.  Output under gpurun_out/<tag>/.
# usage: tools/profile.sh <tag>
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --warmup 2 --cpu-cycles 0 --no-solve --no-fast ${BENCH_EXTRA:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 10 > $out/bench_stats.json 2> $out/stats.err || { echo "stats pass failed"; tail -5 $out/stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B --steps 3 > $out/bench_fetch.json 2> $out/fetch.err || { echo "fetch pass failed"; tail -5 $out/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B --steps 3 > $out/bench_write.json 2> $out/write.err || { echo "write pass failed"; tail -5 $out/write.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rd -- $B --steps 3 > $out/bench_rd.json 2> $out/rd.err || { echo "rd pass failed"; tail -5 $out/rd.err; exit 1; }
echo "profile $tag done"
