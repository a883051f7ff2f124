cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for f in 1 0; do
  rm -rf gpurun_out/prof_train_$f; mkdir -p gpurun_out/prof_train_$f
  MGN_TRAIN_F16=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$f -- python3 tools/step_timing.py > gpurun_out/prof_train_$f.log 2>&1
  f2=$(ls gpurun_out/prof_train_$f/*/*kernel_stats.csv | head -1); echo "== f16=$f"; head -14 $f2 | cut -c1-150
  find gpurun_out/prof_train_$f -name "*kernel_trace.csv" -delete
done
