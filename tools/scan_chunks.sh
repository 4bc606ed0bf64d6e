#!/bin/bash
# usage: tools/scan_chunks.sh ENVVAR v1 v2 ...  -> per-grid kernel times for each value
export TMPDIR=/tmp
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  rm -rf gpurun_out/scan_tmp
  env $var=$v timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/scan_tmp -- python3 bench.py --steps 5 --warmup 1 --cpu-cycles 0 --no-solve > /tmp/scan.json 2>/tmp/scan.err || { tail -3 /tmp/scan.err; exit 1; }
  python3 tools/trace_table.py gpurun_out/scan_tmp > /tmp/scan.txt; grep "k_sweep_rb\|k_residual_restrict" /tmp/scan.txt
done
rm -rf gpurun_out/scan_tmp
