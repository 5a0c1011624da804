#!/bin/bash
# Diagnostic builds of the software-pipelined attention kernel: tools/attn_sp_ablate.sh <name> <-D flags ...> -> gpurun_lib/libattn_<name>.so (timed / checked under MLSD_LIB_PATH).
# SP_ABL bits (1 no exp, 2 no MFMA, 4 no fragment reads, 8 no maximum / rescale) give WRONG results by construction: timing only.
cd "$(dirname "$0")/../mlimgsynth_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -I hip -I ../../include "$@" -c hip/attention.hip -o /tmp/attn_$name.o || exit 1
objs=$(ls ../lib/obj/*.o | grep -v hip_attention.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -lm -lpthread -ldl -o ../../gpurun_lib/libattn_$name.so $objs /tmp/attn_$name.o
