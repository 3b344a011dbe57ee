# same-box A/B of several environment switches against the default: bash scripts/ab2.sh "VAR1=a" "VAR2=b" ...
run() { env $1 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', j['value'], j['roofline']['achieved'], j['config']['loss'])"; }
for i in 1 2; do
  run "ASIS_NOP=1"
  for v in "$@"; do run "$v"; done
done
