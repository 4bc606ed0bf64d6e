#!/bin/bash
# Times the benchmark's 513^3 V-cycle with several builds of the library side by side on one box.
# usage: tools/ab_lib.sh build/a.so build/b.so ...   ("-" = the in-tree build)
for v in "$@"; do
  if [ "$v" = "-" ]; then unset MG_HIP_LIB; else export MG_HIP_LIB=$PWD/$v; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-solve > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/ab.json'))
print("%-16s ms/step %.4f" % (sys.argv[1], d["ms_per_step"]), " ".join("%s=%.1f" % (k, v["avg_us"] * v["launches"] / d["steps"]) for k, v in d["kernels"].items()))
PY
done
