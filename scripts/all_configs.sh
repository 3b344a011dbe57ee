# every BASELINE configuration and training mode, one line each (progress goes to gpurun_out/all_configs.log as it runs)
mkdir -p gpurun_out; : > gpurun_out/all_configs.log
for a in "" "--all-reference-calls" "--config 2" "--config 4" "--config 5" "--train-adapters" "--train-adapters --train-encoder" "--operand bf16"; do
  python bench.py $a --steps 8 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$a', j['value'], j['ms_per_step'], j['roofline']['achieved'], 'precise_level', j['config'].get('precise_level'), 'split_attn_out', j['config'].get('split_attn_out'))" | tee -a gpurun_out/all_configs.log
done
