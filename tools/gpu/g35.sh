R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4s; mkdir -p $O; cd $R; export PYTHONPATH=$R
TRACE_DBG=0,1 timeout 120 python3 tools/gemm_trace.py 8192 10240 1280 17 geglu 2>&1 | grep -v amdgpu | tee $O/r4_gemm_trace_geglu.txt
TRACE_DBG=0,1 timeout 120 python3 tools/gemm_trace.py 8192 10240 1280 17 f16 2>&1 | grep -v amdgpu | tee -a $O/r4_gemm_trace_geglu.txt
