R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 300 python3 tools/mfma_rate.py 40000 12 2>&1 | grep -v amdgpu | tee $O/r4_mfma_rate.txt
MLSD_PROBE_ONE_WAVE=1 timeout 300 python3 tools/mfma_rate.py 40000 12 2>&1 | grep -v amdgpu | tee -a $O/r4_mfma_rate.txt
