cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err
rm -rf gpurun_out/prof_train_cyl; mkdir -p gpurun_out/prof_train_cyl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_cyl -- python3 tools/step_timing.py > gpurun_out/prof_train_cyl.log 2>&1
find gpurun_out/prof_train_cyl -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_train1m_pmc/pmc_$c; mkdir -p gpurun_out/prof_train1m_pmc/pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/prof_train1m_pmc/pmc_$c -- python3 tools/step_1m.py > gpurun_out/prof_train1m_pmc_$c.log 2>&1
done
python tools/pmc_summary.py gpurun_out/prof_train1m_pmc > gpurun_out/pmc_summary_train_step_1m.json
find gpurun_out/prof_train1m_pmc -name "*counter_collection.csv" -size +20M -delete
tail -c 600 gpurun_out/r05_bench_final.json
