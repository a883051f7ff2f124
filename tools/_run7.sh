cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ws
rm -rf $OUT; mkdir -p $OUT
export MGN_FP32_SPLIT=3
CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py $OUT > gpurun_out/pmc_ws.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/pmc_ws.json'))
for k,v in d.items():
    if 'k_edge_ws' in k or 'k_node_step' in k:
        print(k[:60]); 
        for c,x in v.items():
            if c!='derived': print('   ',c, '%.4g'%x['avg_per_launch'])
        print('   derived', v['derived'])
PY
