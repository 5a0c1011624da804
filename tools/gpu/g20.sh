R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 2000 python3 tools/tune_inplan.py $O/tune_r4c.inc unet:sd1:64:2 unet:sdxl:128:8 unet:sdxl:128:16 vae:sdxl:128:4 vae:sd1:64:1 tae:sdxl:128:4 vae:sdxl:128:8 > $O/tune.log 2>&1; grep -E "^(unet|vae|tae):|written|tune_inplan\]" $O/tune.log | tail -30
