#!/bin/bash
# Regenerate the committed profile summaries of one workload on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <workload> <batch> <tag>      e.g.  tools/profile_round.sh sdxl 4 r2
# Writes under gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/.
set -u
WL=${1:-sdxl}; B=${2:-4}; TAG=${3:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
ARGS="--workload $WL --batch-per-gpu $B --steps 1 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$WL -- python3 $R/bench.py $ARGS > $OUT/kt_$WL.log 2>&1 < /dev/null
for f in $(find $OUT/kt_$WL -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${TAG}_${WL}_b${B}_rocprofv3_kernel_stats.csv; done
find $OUT/kt_$WL -name "*kernel_trace.csv" -delete
# the same statistics for the UNet plan ALONE (tools/unet_eval.py: 10 evaluations of the plan bench.py times, nothing else in the process): the per-kernel averages that
# bench.py's roofline.avg_launch_us must agree with (the 140 launches of 'gemm<256x256x64pp,linear>' are TWO instantiations here: GEGLU epilogue (60 per evaluation) and fp16 (80))
case $WL in sdxl) UA0="sdxl 128 $((2*B)) 10";; sd15) UA0="sd1 64 $((2*B)) 10";; *) UA0="$WL 8 $((2*B)) 10";; esac
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktu_$WL -- python3 $R/tools/unet_eval.py $UA0 > $OUT/ktu_$WL.log 2>&1 < /dev/null
for f in $(find $OUT/ktu_$WL -name "*kernel_stats.csv" | head -1); do cp $f $OUT/${TAG}_${WL}_b${B}_unet_eval_rocprofv3_kernel_stats.csv; done
find $OUT/ktu_$WL -name "*kernel_trace.csv" -delete
# HBM-side counters: one pass each, with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass)
# Target: tools/unet_eval.py = the same UNet plan (same shapes, same tuned kernels) evaluated 10x on the NULL stream;
# (a WRITE_SIZE pass occasionally hangs at start-up on this pool: hence the short timeout -- re-run the pass, the summary
# is only written when both passes produced counters)
# rocprofv3's counter service segfaults at the first launch on the engine's own HIP stream when bench.py itself is
# the target (ROCm 7.2), and bench.py's roofline is per launch of a UNet evaluation anyway.
case $WL in sdxl) UARGS="sdxl 128 $((2*B)) 10";; sd15) UARGS="sd1 64 $((2*B)) 10";; *) UARGS="$WL 8 $((2*B)) 10";; esac
if [ "${SKIP_KT:-0}" = "1" ]; then :; fi
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${C}_$WL -- python3 $R/tools/unet_eval.py $UARGS > $OUT/pmc_${C}_$WL.log 2>&1 < /dev/null
done
F=$(find $OUT/pmc_FETCH_SIZE_$WL -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_WRITE_SIZE_$WL -name "*counter_collection.csv" | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then python3 $R/tools/pmc_summary.py $OUT/${TAG}_${WL}_b${B}_pmc_traffic.json "$F" "$W"; fi
# matrix-pipe utilisation (one SQ pass)
timeout 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/pmc_mfma_$WL -- python3 $R/tools/unet_eval.py $UARGS > $OUT/pmc_mfma_$WL.log 2>&1 < /dev/null
for f in $(find $OUT/pmc_mfma_$WL -name "*counter_collection.csv" | head -1); do python3 $R/tools/pmc_mfma_summary.py $OUT/${TAG}_${WL}_b${B}_pmc_mfma.json "$f"; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cd $R
timeout 600 python3 bench.py --workload $WL --batch-per-gpu $B --kernel-table $OUT/${TAG}_${WL}_b${B}_unet_eval_kernel_table.txt > $OUT/${TAG}_${WL}_b${B}_bench.json 2> $OUT/bench_$WL.log < /dev/null
tail -1 $OUT/${TAG}_${WL}_b${B}_bench.json
# per-SHAPE tables (UNet evaluation and VAE / TAE decode) and, for sdxl, the phase stamps of the single-round GEMMs
case $WL in sdxl) M=sdxl; LAT=128;; sd15) M=sd1; LAT=64;; *) M=$WL; LAT=8;; esac
timeout 300 python3 tools/shape_table.py $M $LAT $((2*B)) unet > $OUT/${TAG}_${WL}_b${B}_unet_eval_shape_table.txt 2>> $OUT/bench_$WL.log
timeout 300 python3 tools/shape_table.py $M $LAT $B vae > $OUT/${TAG}_${WL}_b${B}_vae_decode_shape_table.txt 2>> $OUT/bench_$WL.log
timeout 300 python3 tools/shape_table.py $M $LAT $B tae > $OUT/${TAG}_${WL}_b${B}_tae_decode_shape_table.txt 2>> $OUT/bench_$WL.log
if [ "$WL" = "sdxl" ]; then
  for f in f16 f32 f32+res; do timeout 120 python3 tools/gemm_trace.py 8192 1280 1280 18 $f; done > $OUT/${TAG}_gemm_trace_8192x1280x1280.txt 2>> $OUT/bench_$WL.log
  timeout 120 python3 tools/gemm_trace.py 8192 3840 1280 17 f16 > $OUT/${TAG}_gemm_trace_8192x3840x1280.txt 2>> $OUT/bench_$WL.log
  timeout 300 python3 tools/gemm_ksweep.py 8192 1280 > $OUT/${TAG}_gemm_ksweep_8192x1280.txt 2>> $OUT/bench_$WL.log
  timeout 120 python3 tools/attn_bench.py 20 > $OUT/${TAG}_attention_variants.txt 2>> $OUT/bench_$WL.log
  # BASELINE.json configs[4]: TAE decode instead of the KL-VAE
  timeout 600 python3 bench.py --workload $WL --batch-per-gpu $B --tae --no-cpu-baseline > $OUT/${TAG}_${WL}_b${B}_tae_bench.json 2>> $OUT/bench_$WL.log < /dev/null
fi
