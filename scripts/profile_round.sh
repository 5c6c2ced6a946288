#!/bin/bash
# Runs on the GPU box (via gpurun): headline bench, rocprofv3 kernel trace of the same command, then the PMC passes
# (each its own run, one counter block family per pass).  Outputs under gpurun_out/; summarise with
# scripts/pmc_summarise.py afterwards and copy into profiles/.
set -e
export TMPDIR=/tmp
W=${1:-c3_wide}
O=gpurun_out
mkdir -p $O
rm -rf $O/prof_kt $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 400 python3 -u bench.py --workload $W > $O/bench_$W.log 2> $O/bench_$W.err
tail -1 $O/bench_$W.log | cut -c1-400
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -o kt -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_kt.log 2>&1
echo "kernel trace done"
# the same steps without bench.py's live per-kernel timings, so that the averages are those of the launches inside the
# training step alone (the live timings repeat each kernel back to back a few dozen times)
rm -rf $O/prof_kt_step
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_step -o kt -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-rooflines --no-secondary > $O/prof_kt_step.log 2>&1
echo "kernel trace (step only) done"
B="python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-rooflines --no-secondary"
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
echo "fetch done"
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
echo "write done"
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
echo "sq done"
find $O/prof_kt -name '*kernel_stats.csv' | head -1
