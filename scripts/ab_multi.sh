# same-box A/B of several environment variants, interleaved: bash scripts/ab_multi.sh ROUNDS "A=1" "B=2 C=3" ... ("-" = default)
# optional BENCH_ARGS="--config 4" in the environment
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    env $e python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-40s' % '$v', j['value'], 'img/s', j['ms_per_step'], 'ms  dense', j['roofline']['achieved'], 'TF/s')"
  done
done
