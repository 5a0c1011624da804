R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 1500 python3 tools/full_size_latent_parity.py sd1 64 20 2>&1 | grep -v amdgpu | tail -4 | tee $O/r4_full_size_latent_parity.txt
timeout 3300 python3 tools/full_size_latent_parity.py sdxl 128 20 2>&1 | grep -v amdgpu | tail -4 | tee -a $O/r4_full_size_latent_parity.txt
