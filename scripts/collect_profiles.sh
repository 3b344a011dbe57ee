# Round evidence on ONE box (bash scripts/collect_profiles.sh A|B): everything lands under gpurun_out/r05/, the summaries are
# copied into profiles/ afterwards (scripts/kernel_stats_from_db.py, scripts/pmc_traffic.py).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05; mkdir -p $O
if [ "$1" = "A" ]; then
  echo "[A1] bench line (default command)"; python bench.py > $O/bench_line.json 2> $O/bench_line.err && tail -c 600 $O/bench_line.json &&
  echo "[A2] kernel trace, default command (side streams on: overlapped launches are stretched)" &&
  rocprofv3 --kernel-trace --stats -d $O/prof_default -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/prof_default.log 2>&1 &&
  echo "[A3] kernel trace, --single-stream (what the roofline pass times)" &&
  rocprofv3 --kernel-trace --stats -d $O/prof_single -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --single-stream > $O/prof_single.log 2>&1 &&
  echo "[A4] PMC FETCH_SIZE" &&
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --single-stream > $O/pmc_fetch.log 2>&1 &&
  echo "[A5] PMC WRITE_SIZE" &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --single-stream > $O/pmc_write.log 2>&1 &&
  echo "[A5b] PMC matrix-pipe busy cycles + GRBM_GUI_ACTIVE (effective clock)" &&
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --single-stream > $O/pmc_mfma.log 2>&1 &&
  python scripts/mfma_util.py $O/pmc_mfma $O/mfma_util.txt > /dev/null &&
  echo "[A5c] PMC SQ wave-cycle breakdown (issuing / issue-stalled / parked)" &&
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-timing --single-stream > $O/pmc_sq.log 2>&1 &&
  python scripts/sq_counters.py $O/pmc_sq $O/sq_counters.txt > /dev/null &&
  echo "[A6] per-shape GEMM table" &&
  ASIS_BENCH_SHAPES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/shapes.json 2> $O/shapes.txt && grep -c "TF/s" $O/shapes.txt &&
  echo "[A done]"
else
  echo "[B1] config 4 kernel trace (--single-stream)" &&
  rocprofv3 --kernel-trace --stats -d $O/prof_c4 -- python3 bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --single-stream > $O/prof_c4.log 2>&1 &&
  echo "[B2] every configuration" && bash scripts/all_configs.sh && cp gpurun_out/all_configs.log $O/all_configs.log &&
  echo "[B3] hipBLASLt yardstick" && python scripts/blaslt_ref.py > $O/hipblaslt_ref.txt 2>&1 && tail -6 $O/hipblaslt_ref.txt &&
  echo "[B4] persistent vs one-tile-per-workgroup GEMM, interleaved" && python scripts/gemm_p8_probe.py time 9 > $O/gemm_p8_ab.txt 2>&1 && cat $O/gemm_p8_ab.txt &&
  echo "[B done]"
fi
