cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh sd15 1 r5 2>&1 | tail -2
python3 tools/ab_eval.py gpurun_lib/libmlimgsynth_amd_r4.so mlimgsynth_amd/lib/libmlimgsynth_amd.so 2 > gpurun_out/profiles_r5/r5_ab_r4_vs_r5.txt 2>&1
tail -9 gpurun_out/profiles_r5/r5_ab_r4_vs_r5.txt
