R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R; export PYTHONPATH=$R
MLSD_LIB_PATH=$R/gpurun_lib/libmlimgsynth_amd_exp.so timeout 1500 python3 -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | tee $O/pytest_experiments2.txt
