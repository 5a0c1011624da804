R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4p; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 2000 python3 tools/tune_inplan.py $O/tune_r4b.inc unet:sdxl:128:2 unet:sdxl:128:4 unet:sd1:64:1 vae:sdxl:128:1 vae:sdxl:128:2 tae:sdxl:128:4 tae:sd1:64:1 vae:sd1:64:2 vae:sd1:64:4 > $O/tune.log 2>&1; grep -E "^(unet|vae|tae):|written" $O/tune.log
