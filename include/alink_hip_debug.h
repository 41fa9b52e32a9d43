/* alink_hip_debug.h — the A/B and diagnostic switches libalink_hip.so exports BESIDE the product ABI (include/alink_hip.h).
 *
 * Nothing in the reference maps to these (it has no native code: SURVEY.md §0); they exist so that a measurement can run the
 * same binary with one kernel form swapped for another (bench.py's A/B flags, tools/layer_profile.py, tests that assert two
 * forms bit-identical) and so that tools/ can read in-kernel stamps.  Contract:
 *   - PROCESS-GLOBAL and NOT thread-safe: a switch changes which kernel form every handle of the process launches from the
 *     next call on (a few are read once, at alink_backbone_create: noted below).  Set them before the work they are meant
 *     for, from one thread, and set them back.
 *   - No switch changes a RESULT beyond what the product documents: every alternative form is bit-identical to the default
 *     (asserted in tests/test_gpu_backbone.py, tests/test_gpu_conv.py) except where noted.
 *   - A product build may drop them (`make DEBUG_SWITCHES=0` is not implemented: they cost one global each); product code in
 *     a-link_amd/ calls none of them outside tests, tools and bench.py's A/B flags.
 * tests/test_abi.py fails on any exported `alink_*` symbol that neither this header nor alink_hip.h declares. */
#ifndef ALINK_HIP_DEBUG_H
#define ALINK_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- which convolution kernel a layer takes (csrc/backbone.hip's dispatch) --------------------------------------------- */
/* bit mask of widths on the linear-tile kernel (bit0 56, bit1 28, bit2 14, bit3 7 wide maps); default all; a width whose
 * bit is clear takes conv3x3_direct_kernel.  Cannot enable the kernel on a device whose LDS out-of-range probe failed. */
void alink_debug_set_linear(int mode);
/* 1: the linear-tile kernel with its run-time-flag epilogue everywhere (default 0: compile-time epilogues where they exist) */
void alink_debug_set_generic_epilogue(int on);
/* 0: never conv3x3_direct_kernel (everything it would take goes to the implicit GEMM) */
void alink_debug_set_direct(int on);
/* 0: conv3x3_direct_kernel's one-workgroup-per-CU variants only (default 1: the two-per-CU variants where they exist) */
void alink_debug_set_pair(int on);
/* 0: the 112-wide 64 -> 64 layer on the tile kernels instead of the rolling-row kernel (conv3x3_c64.hip) */
void alink_debug_set_c64(int on);
/* 0: stage1_unit1's stride-2 conv2 (+ shortcut) on the implicit GEMM instead of the direct stride-2 kernel (conv3x3_s2c64.hip) */
void alink_debug_set_s2direct(int on);
/* 0: stem and stage1_unit1 conv1 as two launches instead of the fused front kernel (front_c64.hip) */
void alink_debug_set_fuse_stem(int on);
/* 0: projection shortcuts as launches of their own instead of extra K-steps of conv2.  READ AT alink_backbone_create. */
void alink_debug_set_fuse_shortcut(int on);
/* 0: the implicit GEMM stages its tiles through registers (global_load -> ds_write) instead of LDS-DMA */
void alink_debug_set_dma(int on);
/* largest 128-channel grid (workgroups) that still takes the 64-channel form of the linear-tile kernel (default 384) */
void alink_debug_set_fine_max(int n);
/* 0: the 64- / 128-channel form chosen per shard, blind to the call's other shards */
void alink_debug_set_sibling_aware(int on);
/* launches of at most this many output pixels take the latency form (conv3x3_lat.hip); 0 = never; default 1600 */
void alink_debug_set_latency_form(int max_pixels);
/* latency form's block shape: -1 by size (default), 0 = 16 x 16 per wave, 1 = 32 x 16, 2 = 32 x 32 */
void alink_debug_set_latency_tiles(int form);
/* start delay (x 1024 cycles) of the second workgroup on a CU in the linear-tile kernel (default 0; measured out) */
void alink_debug_set_stagger(int n);
/* in-call shards start one after the other's front (default 0; measured out) */
void alink_debug_set_shard_stagger(int on);

/* ---- the pair head (csrc/head.hip) ------------------------------------------------------------------------------------ */
/* 0: batches <= 32 on the generic train chain instead of the three-launch tiny step */
void alink_debug_set_tiny_step(int on);
/* 0: alink_head_train_step_input_grads on the generic chain (train step, then input gradients) for SmallRes' head shape too */
void alink_debug_set_mini_step(int on);
/* 0: the bf16 compute mode's predict on the f32-input kernel with in-flight rounding instead of head_fwd_bf16_kernel */
void alink_debug_set_head_bf16_mfma(int on);

/* ---- SmallRes (csrc/smallres.hip) -------------------------------------------------------------------------------------- */
/* 0: the weight gradients of a train step on the caller's stream, after their dz, instead of a side stream beside the dz chain */
void alink_debug_set_smallres_overlap(int on);
/* 0: a train step ends with two update launches (the tower's, the head's) instead of one (the same bits) */
void alink_debug_set_smallres_one_update(int on);

/* ---- diagnostics ------------------------------------------------------------------------------------------------------ */
/* alink_embed returns after this many convolution launches (0 = the whole chain): per-layer timing by difference.
 * CHANGES THE RESULT (the embedding is garbage): tools only. */
void alink_debug_set_stop_after(int n);
/* alink_embed_profile launches every kernel this many times back to back between its two events (default 4) */
void alink_debug_set_profile_reps(int n);
/* ablation builds of the rolling-row kernels (1 = no stores, 2 = no MFMA, ...: tools/experiments).  CHANGES THE RESULT. */
void alink_debug_set_ablate(int a);
/* device buffer (8 x u64 per workgroup) that the STAMP builds of the tile kernels fill with s_memtime readings; NULL = off */
void alink_debug_set_stamps(void* dev_u64);
/* the hardware contract the linear-tile kernel rests on: a DS read beyond the workgroup's LDS allocation returns zero.
 * Writes 516 probe values to dev_out516 (tests/test_gpu_conv.py reads them); returns ALINK_OK or an error code. */
int alink_debug_lds_oob_probe(float* dev_out516, void* stream);
/* 1 if that probe passed on the current device at alink_init (the linear-tile kernel is used only then) */
int alink_debug_linear_contract_ok(void);

#ifdef __cplusplus
}
#endif
#endif /* ALINK_HIP_DEBUG_H */
