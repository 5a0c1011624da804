R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4j; mkdir -p $O
cd $R; export PYTHONPATH=$R
for d in 0 16 32 48; do echo "== SKINNY_DBG=$d"; SKINNY_DBG=$d timeout 300 python3 tools/gemm_skinny.py 2>&1 | grep -v amdgpu.ids | head -24; done | tee $O/gemm_skinny_dbg.txt
