#!/bin/bash
# experiments on one rank's share: env settings x cases, cycle time of the C++ rank without transport
out=gpurun_out/${1:-exp}; mkdir -p $out
shift
for setting in "$@"; do
  echo "== $setting" >> $out/log.txt
  env $setting timeout -k 10 120 python tools/slab_rank_time.py --cases ${CASES:-8:3} >> $out/log.txt 2>&1 || { echo failed >> $out/log.txt; break; }
done
cat $out/log.txt
