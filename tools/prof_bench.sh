#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box through gpurun):
#   stats      --kernel-trace --stats                  per-kernel durations
#   pmc_fetch  --pmc FETCH_SIZE                         HBM read side   (separate pass: the two do not fit one)
#   pmc_write  --pmc WRITE_SIZE                         HBM write side
#   pmc_sq     SQ / GRBM counters                       issue rates, waves
#   pmc_tcc    TCC_HIT_sum TCC_MISS_sum                 L2 hit rate
# then tools/parse_prof.py turns gpurun_out/prof into profiles/<tag>_*.  The program itself follows `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
P=gpurun_out/prof
rm -rf $P && mkdir -p $P
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $ARGS > $P/stats_run.txt 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 $ARGS > $P/fetch_run.txt 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 $ARGS > $P/write_run.txt 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $P/pmc_sq -- python3 $ARGS > $P/sq_run.txt 2>&1 &&
timeout -k 10 240 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $P/pmc_tcc -- python3 $ARGS > $P/tcc_run.txt 2>&1 &&
python3 tools/parse_prof.py $P gpurun_out/profiles_new "${1:-r01}" "python3 $ARGS" > $P/summary.txt 2>&1
echo "rc=$?"; tail -3 $P/summary.txt; ls gpurun_out/profiles_new
