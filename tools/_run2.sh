cd $GRAFT_REPO_ROOT
python tools/ab.py default noslp --rounds 2 > gpurun_out/ab_noslp_f32.txt 2>&1
MGN_FP32_SPLIT=1 python tools/ab.py default noslp --rounds 2 > gpurun_out/ab_noslp_split.txt 2>&1
python tools/ab.py default noslp --rounds 2 --dtype bf16 > gpurun_out/ab_noslp_bf16.txt 2>&1
tail -3 gpurun_out/ab_noslp_*.txt
