#!/bin/bash
# regenerates the artefacts under profiles/ (run on the GPU box through gpurun; outputs land in gpurun_out/)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -c 600 gpurun_out/bench_n1.json
rm -rf /tmp/prof_bench /tmp/pmc_fetch /tmp/pmc_write
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-extras --no-cpu-baseline --steps 1600 --warmup 160 > /dev/null 2>&1)
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) gpurun_out/bench_kernel_stats.csv
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 "$GRAFT_REPO_ROOT/tools/pmc_probe.py" > /dev/null 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 "$GRAFT_REPO_ROOT/tools/pmc_probe.py" > /dev/null 2>&1)
python3 tools/pmc_summarize.py /tmp/pmc_fetch /tmp/pmc_write gpurun_out/r01 > /dev/null && ls gpurun_out | head -20
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_batched -o b -- python3 "$GRAFT_REPO_ROOT/tools/bench_batched.py" 16,64 > /dev/null 2>&1)
cp $(find /tmp/prof_batched -name "*kernel_stats.csv" | head -1) gpurun_out/batched_kernel_stats.csv
head -c 700 gpurun_out/bench_kernel_stats.csv
