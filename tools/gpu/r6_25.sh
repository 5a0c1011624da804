cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
bash tools/profile_round.sh sdxl 4 r6 2>&1 | tail -1 | cut -c1-300
ls gpurun_out/profiles_r6 | grep "^r6_sdxl" | wc -l
