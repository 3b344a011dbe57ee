# same-box A/B of the round-2 late dispatch changes (8-phase M16 form for K >= 1024 GEMMs and for wide convs) against the earlier rule
run() { env $1 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', j['value'], j['roofline']['achieved'])"; }
for i in 1 2; do
  run "ASIS_NOP=1"
  run "ASIS_GEMM_8P=3 ASIS_GEMM_8P_M16=0 ASIS_CONV_8P=0"
done
