#!/bin/bash
# Second wave-cycle pass: which instruction class keeps the waves busy.  usage: tools/profile_sq2.sh <tag>
set -u
tag=${1:-sq2}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --warmup 2 --cpu-cycles 0 --no-solve --steps 3"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $out/sq -- $B > $out/bench_sq.json 2> $out/sq.err || { echo "sq pass failed"; tail -5 $out/sq.err; exit 1; }
echo "sq2 $tag done"
