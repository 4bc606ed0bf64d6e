export TMPDIR=/tmp
for cfg in 0 1 2 3; do
  export MG_RB2_CFG=$cfg
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_$cfg -- python3 bench.py --steps 5 --warmup 2 --cpu-cycles 0 --no-solve > gpurun_out/ab_$cfg.json 2> gpurun_out/ab_$cfg.err || exit 1
  echo "cfg $cfg"; python3 tools/trace_table.py gpurun_out/ab_$cfg | grep rb2
done
