R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
for rep in 1 2; do for hw in 0 64 256 1024; do echo -n "MLSD_GN_FOLD_MAX_HW=$hw: "; MLSD_GN_FOLD_MAX_HW=$hw timeout 200 python3 tools/two_stream_eval.py sd1 64 2 30 8 2>&1 | tail -1 | cut -c1-60; done; done
