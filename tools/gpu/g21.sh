R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 1500 python3 tools/soak_r4.py 2000 300 12 2>&1 | grep -v amdgpu.ids | tee $O/r4_soak.txt | tail -25
