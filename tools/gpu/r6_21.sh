cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
for n in 0 1 2 4 8 16 9 25 27; do
  if [ $n = 0 ]; then L=mlimgsynth_amd/lib/libmlimgsynth_amd.so; else L=gpurun_lib/libattn_abl_$n.so; fi
  echo "== SP_ABL=$n"; MLSD_LIB_PATH=$L SP_SHAPES=2 timeout 300 python3 tools/attn_sp_bench.py 20 2>&1 | grep "software-pipelined  "
done > gpurun_out/r6/attn_sp_ablate2.txt 2>&1
cat gpurun_out/r6/attn_sp_ablate2.txt
