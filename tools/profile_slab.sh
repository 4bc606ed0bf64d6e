#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel statistics, HBM traffic counters (separate passes) and the
# wave-cycle counters for ONE rank's share of the slab-partitioned 1025^3 cycle (no transport:
# tools/slab_rank_time.py).  Output under gpurun_out/<tag>/; condensed by tools/summarize_profile.py.
# usage: [SIZE=2049] tools/profile_slab.sh <tag> [world:rank]
set -u
tag=${1:-slabprof}; case_=${2:-8:3}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
size=${SIZE:-1025}
B="python3 tools/slab_rank_time.py --size $size --cases $case_"
echo "rocprofv3 --kernel-trace --stats -- $B (rank $case_ (world:rank) of the $size^3 slab solve, MG_TRANSPORT_NONE, 2 + 10 cycles); --pmc FETCH_SIZE / WRITE_SIZE / SQ counters in separate runs" > $out/command.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/run_stats.log 2> $out/stats.err || { echo "stats pass failed"; tail -5 $out/stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/run_fetch.log 2> $out/fetch.err || { echo "fetch pass failed"; tail -5 $out/fetch.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/run_write.log 2> $out/write.err || { echo "write pass failed"; tail -5 $out/write.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $out/sq -- $B > $out/run_sq.log 2> $out/sq.err || { echo "sq pass failed"; tail -5 $out/sq.err; exit 1; }
cat $out/run_stats.log | grep world
echo "profile $tag done"
