R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 300 python3 tools/step_phases.py sdxl 2 2>&1 | grep -v amdgpu | tee $O/step_phases.txt
timeout 300 python3 tools/step_phases.py sd15 3 2>&1 | grep -v amdgpu | tee -a $O/step_phases.txt
