cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
OUT=$R/gpurun_out/r6/pmc_valu
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/unet_eval.py sdxl 128 8 4 > $OUT.log 2>&1 < /dev/null
F=$(find $OUT -name "*counter_collection.csv" | head -1)
cd $R
python3 tools/pmc_valu_summary.py gpurun_out/r6/r6_sdxl_b4_pmc_valu.json "$F" | head -14
find $OUT -name "*.csv" -delete
