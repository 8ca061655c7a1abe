#!/bin/bash
# One GPU-box session: parity tests, bench line, kernel-trace stats and the PMC passes
# (VALU / FETCH / WRITE, each alone with --kernel-trace), plus the FETCH/WRITE calibration.
# Usage: tools/gpu_round.sh <tag> [stages...]   stages: test bench stats valu fetch write calib
set -u
TAG=${1:-r03}; shift || true
STAGES=${*:-test bench stats valu fetch write calib}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH_ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-live-pmc --quick-verify"
# the kernel-stats pass runs the driver's step counts, so that its per-kernel averages are over warm launches like the line's
# (with 4 + 1 steps a third of the launches are the first, slower ones of a fresh box: 8.8 against 8.7 ms for the secp256k1 ladder)
STATS_ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --quick-verify"
for s in $STAGES; do
  case $s in
    test)  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest.log ;;
    bench) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --out $OUT/bench_line.json > $OUT/bench_stdout.txt 2>&1; echo "bench rc=$?" | tee -a $OUT/summary.txt ;;
    stats) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/stats -- python $GRAFT_REPO_ROOT/bench.py $STATS_ARGS > $GRAFT_REPO_ROOT/$OUT/stats.log 2>&1); echo "stats rc=$?" | tee -a $OUT/summary.txt ;;
    valu)  (cd /tmp && timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_valu -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/pmc_valu.log 2>&1); echo "valu rc=$?" | tee -a $OUT/summary.txt ;;
    fetch) (cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1); echo "fetch rc=$?" | tee -a $OUT/summary.txt ;;
    write) (cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py $BENCH_ARGS > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1); echo "write rc=$?" | tee -a $OUT/summary.txt ;;
    calib) (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/calib_fetch -- $GRAFT_REPO_ROOT/tools/_build/pmc_calib > $GRAFT_REPO_ROOT/$OUT/calib.log 2>&1;
            timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/calib_write -- $GRAFT_REPO_ROOT/tools/_build/pmc_calib >> $GRAFT_REPO_ROOT/$OUT/calib.log 2>&1); echo "calib rc=$?" | tee -a $OUT/summary.txt ;;
  esac
done
# keep what is small: drop per-dispatch traces larger than a few MB
if ls $OUT/pmc_* >/dev/null 2>&1; then
  python tools/pmc_summary.py $(ls -d $OUT/pmc_* $OUT/calib_* 2>/dev/null | grep -v '\.log') --out $OUT/pmc.json 2>&1 | tail -1
fi
find $OUT -name "*.csv" -size +3M -delete
find $OUT -name "*.db" -delete
du -sh $OUT
cat $OUT/summary.txt
