cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_comm.py -x -q -k "bench" > gpurun_out/t_bench.txt 2>&1
tail -n 12 gpurun_out/t_bench.txt
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.err
python tools/benchline.py gpurun_out/bench_default.json 2>/dev/null | head -80
