#!/bin/bash
# GPU-box script (round 3): counters of EVERY kernel the bench command launches -> gpurun_out/r03_pmc_traffic.json (copied to profiles/).
# Separate rocprofv3 --pmc passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: they cannot share a pass), no trace domains combined with
# them, the program directly after `--`; then the kernel-trace stats of the same command.
#   per (kernel, grid, workgroup): FETCH_SIZE / WRITE_SIZE (KB per launch), GRBM_GUI_ACTIVE (sum over the 8 XCDs),
#   SQ_ACTIVE_INST_VALU (quad-cycles), SQ_INSTS_VALU, SQ_INSTS_LDS, SQ_LDS_BANK_CONFLICT, SQ_WAVES
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-host-threads"
run() { timeout -k 5 400 rocprofv3 --pmc "${@:2}" --output-format csv -d $R/gpurun_out/r03_pmc_bench_$1 -o p -- $CMD > $R/gpurun_out/r03_pmc_bench_$1.log 2>&1 || { tail -5 $R/gpurun_out/r03_pmc_bench_$1.log; exit 1; }; }
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
run sq SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_bench_trace -o t -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-host-threads > $R/gpurun_out/r03_bench_traced.json 2> $R/gpurun_out/r03_bench_traced.err
python3 $R/tools/pmc_traffic_summary.py $R/gpurun_out r03_pmc_bench_ $R/gpurun_out/r03_pmc_traffic.json
cp $R/gpurun_out/r03_bench_trace/t_kernel_stats.csv $R/gpurun_out/r03_bench_kernel_stats.csv
grep -h "svthip" $R/gpurun_out/r03_bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | head -40
cut -c1-300 $R/gpurun_out/r03_bench_traced.json
