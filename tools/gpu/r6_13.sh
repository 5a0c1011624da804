cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
F='^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids'
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "software_pipelined or test_attention" 2>&1 | grep -v "$F" | tail -8 ) > gpurun_out/r6/t13_attn_tests.txt; tail -4 gpurun_out/r6/t13_attn_tests.txt
rocm-smi --showpower --showclocks --showuse --csv 2>&1 | head -5
rocm-smi --showmaxpower 2>&1 | grep -i "max\|power" | head -3
bash tools/power_watch.sh gpurun_out/r6/power_sdxl_b4_eval.txt -- timeout 300 python3 tools/unet_eval.py sdxl 128 8 150 2>&1 | tail -1
bash tools/power_watch.sh gpurun_out/r6/power_mfma_rate.txt -- timeout 300 python3 tools/mfma_rate.py 40000 40 2>&1 | grep -v "$F" | tail -3
bash tools/power_watch.sh gpurun_out/r6/power_sd15_b1_eval.txt -- timeout 300 python3 tools/unet_eval.py sd1 64 2 1500 2>&1 | tail -1
for f in power_sdxl_b4_eval power_mfma_rate power_sd15_b1_eval; do echo "== $f"; sed -n '8p;16p;24p;32p' gpurun_out/r6/$f.txt; wc -l gpurun_out/r6/$f.txt; done
