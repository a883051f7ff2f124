cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_fp32_split.py tests/test_gpu_two_edge_sets.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
for f in 1; do
  rm -rf gpurun_out/prof_small_$f; mkdir -p gpurun_out/prof_small_$f
  MGN_SPLIT_F16=$f timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_small_$f -- python3 tools/small_mesh_loop.py 40 > gpurun_out/prof_small_$f.log 2>&1
  f2=$(ls gpurun_out/prof_small_$f/*/*kernel_stats.csv | head -1); echo "== f16=$f"; head -4 $f2 | cut -c1-150
  find gpurun_out/prof_small_$f -name "*kernel_trace.csv" -delete
done
