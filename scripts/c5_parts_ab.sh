# config 5 (ViT-g/14 40 SwiGLU blocks + MLA, 11 classes) at precise_level 2: which linear layers need split operands?
# For every subset of interest: the full-depth stress golden (tests/test_gpu_fulldepth.py, B = 1) and bench.py --config 5 (B = 12).
#   bash scripts/c5_parts_ab.sh > gpurun_out/c5_parts_ab.txt
set -o pipefail
for parts in "qkv,proj,fc1,fc2" "qkv,proj,fc1" "proj,fc1" "qkv,proj" "proj,fc1,fc2" "proj"; do
  echo "== ASIS_PRECISE_PARTS=$parts"
  ASIS_PRECISE_PARTS=$parts python -m pytest tests/test_gpu_fulldepth.py -x -q -s -k "config5_vitg and kernel" 2>&1 | grep -E "output \(11 classes|passed|failed|AssertionError" | head -4
  ASIS_PRECISE_PARTS=$parts python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench config 5:', d['value'], 'img/s', d['ms_per_step'], 'ms/step')"
done
