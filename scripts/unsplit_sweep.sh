# which conv layers need split-precision operands?  logits rel-L2 of the ViT-L/14 588^2 step goldens per unsplit set
# (ASIS_ENC_STREAM=0: one stream, so ASIS_UNSPLIT is the only difference between the runs)
for u in ${1:-"" d1 d2 d3 d4 d3,d4 d2,d3,d4 d1,d2,d3,d4 stem3,stem6,conv2,conv3,conv4}; do
  echo "== ASIS_UNSPLIT=$u"
  ASIS_UNSPLIT=$u python -m pytest tests/test_gpu_step.py -q -m gpu -s -k golden 2>&1 | grep "step_exact {\|step_kernel {" | sed -e "s/'x_final'.*'logits'/'logits'/" | cut -c1-120
done
