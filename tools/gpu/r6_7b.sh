cd $GRAFT_REPO_ROOT
python3 tools/ln_dbg_tmp.py 2>&1 | grep "max err\|gemm<"
bash tools/gpu/r6_7.sh
