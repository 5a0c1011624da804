#!/bin/bash
# CPU-only memory-safety check of the C host side (the GPU pool has no sanitizer runs): builds the host sources with
# AddressSanitizer against the already built HIP objects, runs the CPU tests that exercise host logic (dry runtime: plan builder,
# loaders, public API, prompt parser, tokenizer) and the loader fuzz with that library, then restores the normal one.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); B=/tmp/mlsd_asan; mkdir -p $B
# the command line's own readers (PNG / inflate / PNM / tensor files): an AddressSanitizer build of the CLI against the same library
(cd $R && gcc -O1 -g -std=gnu11 -fsanitize=address -fno-omit-frame-pointer -I include -o $B/mlimgsynth-amd-asan mlimgsynth_amd/csrc/cli/main.c -L mlimgsynth_amd/lib -lmlimgsynth_amd -Wl,-rpath,$R/mlimgsynth_amd/lib -Wl,-rpath,/opt/rocm/lib -lm)
(cd $R && ASAN_OPTIONS=detect_leaks=0 python3 tools/fuzz_cli_readers.py $B/mlimgsynth-amd-asan 400)
cd $R/mlimgsynth_amd/csrc
for f in host/*.c; do gcc -O1 -g -std=gnu11 -fPIC -fsanitize=address -fno-omit-frame-pointer -ffp-contract=off -I host -I ../../include -c $f -o $B/$(basename $f .c).o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -Wl,-rpath,/opt/rocm/lib -lm -lpthread -ldl -o $B/libmlimgsynth_amd.so ../lib/obj/hip_*.o $B/*.o
cp ../lib/libmlimgsynth_amd.so $B/orig.so; trap "cp $B/orig.so $R/mlimgsynth_amd/lib/libmlimgsynth_amd.so" EXIT
cp $B/libmlimgsynth_amd.so ../lib/libmlimgsynth_amd.so
cd $R
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python3 -m pytest tests/test_host_cpu.py tests/test_loader_cpu.py tests/test_api_cpu.py tests/test_prompt_cpu.py tests/test_tokenizer_cpu.py -x -q
python3 tools/fuzz_loaders.py
