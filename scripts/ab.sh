# same-box A/B of an environment switch: bash scripts/ab.sh VAR OFFVALUE
for i in 1 2; do
  env $1=$2 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1=$2', j['value'], j['roofline']['achieved'])"
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('default', j['value'], j['roofline']['achieved'])"
done
