"""Stated parity tolerances of the -m gpu tests (rel-L2 against the oracle / the independent torch vectors).

Round 4 lowered them to what the design supports (SURVEY 8c planned "<= ~2e-3" per evaluation; measured 0.6e-3 .. 1.4e-3):
a regression that doubles the error must fail.  Quoted for the oracle's DEFAULT mode (exact fp32 GELU); the HIP path against the
oracle's ggml-F16-table mode (orc_set_ggml_f16_tables) is reported in profiles/r4_parity_f16_tables.txt and held to the same bound
by tests/test_unet_gpu.py::test_unet_parity_in_the_ggml_f16_table_mode.

Op arithmetic is PARITY-UNPINNED: ggml, where the reference's tensor arithmetic lives, is absent from /root/reference; the bounds are
against a cited CPU restatement (oracle/) corroborated by an independent torch restatement (tools/torch_ref.py)."""
EVAL = 2e-3      # VAE / TAESD decode and encode; CLIP embeddings and features (fp16 GEMM / attention operands, fp32 order); the generic per-pass bound
# One UNet evaluation, by latent size (round 6).  The rel-L2 figure of an evaluation is not a property of the kernels alone: at TEST-SIZED latents (8 .. 32:
# 4 x 4 maps at the lowest level, a few hundred values per GroupNorm group) it scatters by +-15 % with the fp32 SUMMATION ORDER of equally exact kernels -- the same SDXL evaluation
# at a 16 x 16 latent measured 1.71e-3 with the static tile rule, 2.07e-3 with other K splits (every candidate GEMM within 1-7e-7 of float64, tools/splitk_accuracy.py), 2.23e-3 with
# the LayerNorms in their own launches, 2.18e-3 with two-pass GroupNorm statistics -- while the evaluations at BASELINE's sizes (SDXL 128 x 128, SD1.5 64 x 64
# latents; batch 1, 2, 8, 16 plans, streamed or resident) measure 1.0 - 1.2e-3.  One bound for both misjudges both: a harmless K-split change at 16 x 16 trips 2e-3, and a real 40 %
# regression at 128 x 128 passes it.  Hence two stated bounds:
EVAL_HEADLINE = 1.6e-3   # UNet evaluations at BASELINE's sizes (measured 1.0 - 1.2e-3: a 40 % regression fails)
EVAL_SMALL = 2.5e-3      # UNet evaluations at test-sized latents (measured 0.6 - 2.2e-3 depending on the summation order; the scatter, not a kernel, sets this bound)
LATENT = 1e-2    # final latent of a complete sampled generation (20-step Euler-a and the other solvers: errors of all evaluations compound)
