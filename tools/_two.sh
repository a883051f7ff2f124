cd $GRAFT_REPO_ROOT
MGN_BENCH_ONE_GPU=1 python bench.py --gpus 2 --nx 500 --no-secondary --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r05_two_ranks_one_gpu.json 2> gpurun_out/r05_two_ranks.err
python - <<EOF2
import json
d=json.loads(open("gpurun_out/r05_two_ranks_one_gpu.json").read().strip().splitlines()[-1])
print(d["ms_per_processor_step"], d.get("transport"), d["roofline"].get("halo_pack_ms"), d["per_rank"])
EOF2
tail -3 gpurun_out/r05_two_ranks.err
