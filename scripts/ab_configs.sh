# same-box A/B of one environment switch over every BASELINE configuration: bash scripts/ab_configs.sh VAR=value
for a in "" "--config 2" "--config 4" "--config 5" "--train-adapters"; do
  for e in "ASIS_NOP=1" "$1"; do
    env $e python bench.py $a --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$a', '$e', j['value'], j['roofline']['achieved'], j['config']['loss'])"
  done
done
