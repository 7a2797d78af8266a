#!/bin/bash
# regenerates the artefacts under profiles/ for the current round (run on the GPU box through gpurun; outputs land in
# gpurun_out/ and are copied into profiles/ by hand once checked).  usage: bash tools/refresh_profiles.sh r02
R=${1:-r02}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python3 bench.py > gpurun_out/${R}_bench_n1.json 2> gpurun_out/${R}_bench_n1.err
tail -c 400 gpurun_out/${R}_bench_n1.json
rm -rf /tmp/prof_bench /tmp/prof_pipe /tmp/pmc_fetch /tmp/pmc_write /tmp/prof_cfg /tmp/prof_batched
# headline run (resident kernel) and the same run on the two-launch pipeline
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-extras --no-cpu-baseline --steps 1600 --warmup 160 > /dev/null 2>&1)
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_kernel_stats.csv
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pipe -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --resident 0 --no-extras --no-cpu-baseline --steps 1600 --warmup 160 > /dev/null 2>&1)
cp $(find /tmp/prof_pipe -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_bench_pipeline_kernel_stats.csv
# HBM traffic counters: separate passes, no tracing domains beside them
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$GRAFT_REPO_ROOT/tools/pmc_probe.py" > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$GRAFT_REPO_ROOT/tools/pmc_probe.py" > /dev/null 2>&1)
python3 tools/pmc_summarize.py /tmp/pmc_fetch /tmp/pmc_write gpurun_out/${R} > /dev/null && ls gpurun_out | grep ${R}
# BASELINE configs[1] (FISTA + L1) and configs[2] (ADMM + TV): per-kernel statistics
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg -o c -- python3 "$GRAFT_REPO_ROOT/tools/bench_configs.py" > /dev/null 2>&1)
cp $(find /tmp/prof_cfg -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_config2_config3_kernel_stats.csv
# shared-A batched solves (BASELINE configs[3] on one GPU)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batched -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched.py" 8,16,64 > /dev/null 2>&1)
cp $(find /tmp/prof_batched -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_batched_kernel_stats.csv
# the same on the reference's default operator (AHA explicit, shared by the columns): streaming product and resident launch
rm -rf /tmp/prof_bg /tmp/prof_bgs
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bg -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched.py" 8,16,64 gram=1 > "$GRAFT_REPO_ROOT/gpurun_out/${R}_batched_gram.log" 2>&1)
cp $(find /tmp/prof_bg -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_batched_gram_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bgs -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched.py" 8 gram=1 resident=0 > "$GRAFT_REPO_ROOT/gpurun_out/${R}_batched_gram_streaming.log" 2>&1)
cp $(find /tmp/prof_bgs -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_batched_gram_streaming_kernel_stats.csv
# in-kernel primitives (bare group / grid barriers, the exchanges with the arithmetic stripped), cadence, config 1
tools/ubench/grid_barrier 2000 2048 > gpurun_out/${R}_grid_barrier.txt 2>&1
python3 tools/bench_cadence.py > gpurun_out/${R}_cadence.txt 2>&1
python3 tools/bench_config1.py > gpurun_out/${R}_config1.txt 2>&1
tools/ubench/launch_floor > gpurun_out/${R}_launch_floor.txt 2>&1
python3 tools/bench_small_group.py > gpurun_out/${R}_small_group.txt 2>&1
python3 tools/bench_tv_prox.py > gpurun_out/${R}_tv_prox.txt 2>&1
python3 tools/bench_batched_fista.py > gpurun_out/${R}_batched_fista.txt 2>&1
rm -rf /tmp/prof_bf
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched_fista.py" > /dev/null 2>&1)
cp $(find /tmp/prof_bf -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_batched_fista_kernel_stats.csv
cp gpurun_out/parity_errors.jsonl gpurun_out/${R}_parity_errors.jsonl 2>/dev/null
head -c 900 gpurun_out/${R}_bench_kernel_stats.csv
