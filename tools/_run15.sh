cd $GRAFT_REPO_ROOT
bash profiles/run_profiles.sh r03 > gpurun_out/prof_r03.log 2>&1
MGN_FP32_SPLIT=0 bash profiles/run_profiles.sh r03_fp32 > gpurun_out/prof_r03_fp32.log 2>&1
bash profiles/run_profiles.sh r03_bf16 --dtype bf16 > gpurun_out/prof_r03_bf16.log 2>&1
for t in r03 r03_fp32 r03_bf16; do timeout 60 python3 tools/pmc_summary.py gpurun_out/prof_$t > gpurun_out/pmc_summary_$t.json; find gpurun_out/prof_$t/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_$t.csv; done
ls -la gpurun_out/*.json gpurun_out/*.csv | head; head -8 gpurun_out/kernel_stats_r03.csv
