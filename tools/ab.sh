#!/bin/bash
# usage: tools_ab.sh "ENV=.. ENV2=.." ...   -> runs bench (no cpu, no solve) per variant, prints kernel table
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 120 python bench.py --steps 10 --warmup 2 --cpu-cycles 0 --no-solve > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
  python - <<'PY'
import json
d=json.load(open('/tmp/ab.json'))
print("ms/step %.3f  vps %.1f  GDoF/s %.1f"%(d["ms_per_step"],d["vcycles_per_s"],d["value"]))
for k,v in d["kernels"].items():
    print("  %-16s n=%5d avg_us=%9.1f opeq GB/s=%8.1f share=%.3f"%(k,v["launches"],v["avg_us"],v["operator_equivalent_GBps"],v["share"]))
PY
done
