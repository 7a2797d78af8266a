cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_gpu_suite.txt
bash tools/refresh_profiles.sh r06 > gpurun_out/r06_refresh.log 2>&1; tail -5 gpurun_out/r06_refresh.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_like.json 2> gpurun_out/r06_bench_driver_like.err; tail -c 300 gpurun_out/r06_bench_driver_like.json
python3 tools/stamps_resident.py > gpurun_out/r06_stamps_resident_raw.txt 2>&1; tail -12 gpurun_out/r06_stamps_resident_raw.txt
rm -rf /tmp/pmc_sq; (cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_sq -- python3 "$GRAFT_REPO_ROOT/tools/pmc_probe.py" > /dev/null 2>&1)
python3 tools/pmc_sq_summarize.py /tmp/pmc_sq gpurun_out/r06_pmc_sq.csv | head -12
python3 bench.py --gpus 2 --rehearse --steps 64 --warmup 32 --c5-rows 8192 > gpurun_out/r06_bench_rehearse_2ranks.json 2>/dev/null; tail -c 200 gpurun_out/r06_bench_rehearse_2ranks.json
