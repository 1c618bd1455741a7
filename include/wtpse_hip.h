/* wtpse_hip.h — C ABI of libwtpse_hip.so: the MI355X (gfx950) kernels behind the WT-PSE training hot path.
 *
 * The reference (tonyckc/WT-PSE-code) has no FFI of its own: its boundary is the Python class surface of
 * algorithms.py / shape_networks.py, which wt-pse-code_amd/{algorithms,shape_networks}.py mirror.
 * This library sits directly below that surface.  Each entry point replaces the stock ATen dispatches the
 * reference reaches from the cited lines.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; every pointer is DEVICE memory unless stated; tensors are NCHW fp32,
 *     contiguous, caller-owned.  No allocation, no host synchronisation and no stream creation inside: scratch
 *     ("partial", "slab", "ws") is passed in, `stream` is a hipStream_t (NULL = default stream), so every call is
 *     legal inside hipGraph capture.
 *   - return 0 on success, -1 (WTPSE_EINVAL) for rejected arguments, otherwise the hipError_t of the launch.
 *   - "pro" (prologue) = optional per-channel [C][2] (scale, shift) applied to an input as it is loaded, followed by
 *     ReLU when the matching relu bit is set: BatchNorm-apply + activation fused into the consumer.
 */
#ifndef WTPSE_HIP_H
#define WTPSE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- convolution: nn.Conv2d 3x3 pad 1 / 1x1 (algorithms.py:882-888,926-933,404-424,991,1006-1012,1199-1201;
 *      shape_networks.py:182-193,332-338,376-383,459-465) ------------------------------------------------------- */

/* OIHW parameters -> kernel layouts, all convs of a network in one launch.
 * desc: n_desc x 8 ints {w_off, Cout, Cin, taps, wf_off (-1: none), wd_off (-1: none), 0, 0}, offsets in floats into
 * `params` / `packed`.  wf = [ceil4(Cin)][taps][ceil16(Cout)], wd = [ceil4(Cout)][taps][ceil16(Cin)] (tap-flipped). */
int wtpse_pack_conv_weights(const float* params, const int* desc, int n_desc, float* packed, void* stream);

/* out = conv(cat(in0, in1)) [+ bias] [ReLU].  in1 may be NULL (C1 = 0): torch.cat (algorithms.py:955,1018) is virtual.
 * pro0 / pro1: [C0][2] / [C1][2] (scale, shift) for in0 / in1, or NULL; pro_relu bit0 / bit1: ReLU on in0 / in1 after it.
 * Output channels [0, Csplit) go to out0, the rest to out1 (Csplit == Cout, out1 NULL: no split; otherwise Csplit % 16 == 0).
 * stats: NULL or [wtpse_conv_stats_blocks(B,H,W)][Cout][2] per-workgroup (sum, sum^2) of the output (train-mode
 * BatchNorm statistics, algorithms.py:883-889); not combinable with relu_out.
 * mask_ref: NULL or [B][Cout][H][W]: out = mask_ref > 0 ? value : 0 (the ReLU backward of the layer below, fused into
 * its data gradient; not combinable with a split).
 * out_amax: NULL or the amax table (see wtpse_amax; ZERO on entry) of the stored output — what an x2h consumer scales this tensor by
 * when no train-mode BatchNorm sits in between (see wtpse_x3_terms); not combinable with mask_ref.
 * The data gradient is this same call on dY with the `wd` layout. */
int wtpse_conv_fwd(const float* in0, int C0, const float* in1, int C1, const float* wpacked, const float* bias,
                   const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1, int Csplit, float* stats,
                   int B, int H, int W, int Cout, int ksize, int relu_out, const float* mask_ref, unsigned* out_amax, void* stream);
int wtpse_conv_stats_blocks(int B, int H, int W);
/* 3x3 convolution with exactly 16 output channels (DeepWT, algorithms.py:1091-1117) that also emits the per-tile partial
 * Grams of its output in the epilogue: gram_partial [wtpse_conv_stats_blocks(B,H,W)][256] = the WT loss's `partial` layout
 * with S = tiles per image (wtpse_wt_loss_fwd_partials), so that compute_whitening_loss never re-reads z from HBM.
 * relu_out must be 0 (the Gram describes the stored map: the epilogue works on the pre-activation accumulators). */
int wtpse_conv_fwd_gram(const float* in0, int C0, const float* wpacked, const float* bias, const float* pro0, int pro_relu,
                        float* out0, float* gram_partial, int B, int H, int W, int Cout, int relu_out, unsigned* out_amax,
                        void* stream);

/* The same convolution on the BF16 matrix cores at fp32 accuracy (csrc/conv_x3.hip): every fp32 operand is split into three
 * bf16 terms and the product formed from the six leading cross terms with fp32 accumulation (6 bf16 MFMAs instead of 8 fp32
 * MFMAs per 32x32x16 block).  Same contract as wtpse_conv_fwd; requires Cout > 16 and ceil16(C0 + C1) <= 256 (<= 512 when
 * the launch uses 64-channel row blocks: Cout % 64 == 0 and at least 512 of them) — the per-channel prologue coefficients are
 * staged in LDS; anything else fails with WTPSE_ERR_ARG.  `wpacked`: the weights pre-split by
 * wtpse_pack_conv_weights_x3 — desc as for wtpse_pack_conv_weights with offsets {xf_off, xd_off} in unsigned shorts; per
 * conv and direction 32 + ceil16(K) * ceil32(rows) * taps * 3 unsigned shorts (forward: rows = Cout, K = Cin; data gradient:
 * rows = Cin, K = Cout): a 64-byte header (float {1 / scale, scale, 0, 0, 12 words of the packer's scratch}: the layer's
 * power-of-two weight scale, 1 unless wtpse_x3_terms() == 2), then [K chunk 16][row block 32][tap][term slot 3][k half 2][row 32][8 k]. */
int wtpse_conv_x3_stats_blocks(int B, int H, int W, int Cout, int ksize);   /* rows of `stats` for wtpse_conv_fwd_x3 (the tiling depends on the kernel size) */
/* Which 3x3 launches of the x3 entry points run conv_x3r_k (weights fed from registers, input tile double-buffered in LDS) instead
 * of conv_x3_k (weights staged through LDS): on = 1 (default) the launches with 64-channel output blocks, 2 all of them, 0 none;
 * on < 0 only queries (environment: WTPSE_X3R=0|1|2).  Returns the previous setting.  The two kernels give bitwise the same
 * results (tests/test_conv_x3_gpu.py::test_x3r_equals_x3); the choice is by measurement (profiles/r04_microbench_x3.txt). */
int wtpse_x3r_enable(int on);
/* Order in which the workgroups of the x3 convolutions take their (tile, output-channel block) pairs: on = 1 (default) XCD-aware —
 * the hardware deals consecutive workgroups to the 8 XCDs in turn; each XCD is given a contiguous range of tiles and runs a tile's
 * output-channel blocks back to back, so that a tile's input (and the halo it shares with its neighbours) is fetched into ONE L2
 * once; 0 = dispatch order (tile fastest); on < 0 only queries (environment: WTPSE_X3_XCD=0|1).  Returns the previous setting.
 * Same workgroups, bitwise the same results (tests/test_conv_x3_gpu.py::test_xcd_order_equals_dispatch_order). */
int wtpse_x3_xcd(int on);
/* Tiling of the 32-channel-block launches on maps wider than 16 pixels: on = 1: 128-pixel (32 x 4) tiles — twice as many, half as long
 * workgroups, five resident per CU instead of four — on = 0 (default until measured otherwise): 256-pixel (32 x 8) tiles; on < 0 only
 * queries (environment: WTPSE_X3_SMALL_WIDE=0|1).  Returns the previous setting.  Changes the rows of `stats`
 * (wtpse_conv_x3_stats_blocks follows it) and is part of wtpse_tuning_state(). */
int wtpse_x3_small_wide(int on);
/* Arithmetic of the x3 kernels (wtpse_conv_fwd_x3 and the data gradients on it, wtpse_conv_wgrad_r), by the number of 16-bit terms
 * an fp32 operand is split into:
 *   3 = "x3": three bf16 terms, six MFMA products per multiply (round 2);
 *   2 = "x2h" (default since round 5): TWO fp16 terms (11 + 11 significant bits, each rounded to nearest even, the remainder exact
 *       in fp32), THREE products a0 b1 + a1 b0 + a0 b0 on v_mfma_f32_*_f16, fp32 accumulation — half the MFMAs of x3 at the same
 *       measured accuracy (tests/test_conv_x3_gpu.py: against fp64 beside x3 and the fp32-input MFMA).  fp16 has 5 exponent bits, so
 *       every operand tensor is multiplied by a power of two as it is loaded (exact) and the result scaled back: weights per layer
 *       (from the layer's largest magnitude, found by wtpse_pack_conv_weights_x3); an input tensor by the power of two that brings
 *       `in_amax` — an amax table (wtpse_amax below) holding a bound of the largest magnitude of the input AS LOADED, i.e. after the
 *       prologue — into [2^14, 2^15): full 22-bit precision for every element down to 2^-17 of the bound, an absolute error of 2^-39
 *       of the bound below that.  For a GRADIENT the table is left by the tensor's producer (wtpse_bn_bwd_apply_coef,
 *       wtpse_upsample2x_bwd[_bn], ...) or by wtpse_amax.  For a FORWARD ACTIVATION (round 6) it is left by whoever knows a bound:
 *       the train-mode BatchNorm finalize (wtpse_bn_finalize / wtpse_conv_fwd_bnf, `act_amax`: |gamma| sqrt(N - 1) + |beta| per
 *       channel by Samuelson's inequality — no look at the data), the producing convolution's epilogue (`out_amax`: un-normalised
 *       maps), wtpse_act_bound (an eval-mode BatchNorm or any per-channel affine map over a tensor of known amax) or wtpse_amax.
 *       A concat has one table per half (in_amax / in_amax1: the larger counts).  No table at all (NULL): the fixed scale 2^2 of
 *       round 5 — full precision only for 2^-5 <= |x| < 2^14, NaN beyond: right for O(1) data, a fallback for bare callers only
 *       (wtpse_hip/nn.py always passes tables).  Non-finite values: a NaN or inf operand makes every output it feeds NaN (the fp16
 *       terms of an inf are (inf, NaN): an inf does not stay an inf as it can in fp32); forward launches store a NaN accumulator as
 *       NaN, with or without an output ReLU, as torch's conv / ReLU do (the data gradients' epilogues clamp with v_max_f32, which
 *       turns NaN into the clamp bound: -inf or 0 — non-finite, or masked, either way).  A consumer's ReLU-ON-LOAD is v_max_f32 too and
 *       maps NaN to 0 where torch.relu keeps it: a NaN stays loud through BatchNorm statistics (its whole channel turns NaN) and through
 *       un-activated paths (the heads' outputs, mu, the logits), not through an activation;
 *   1 = the `bf16` mode of BASELINE.json configs[1]: operands rounded to ONE bf16 term, one product — outside the 1e-4 parity bar by
 *       construction (tests/test_bf16_mode_gpu.py states its tolerance).
 * The weight gradients of the 16-pixel-wide maps (wtpse_conv_wgrad_x3) stay on x3.  Packed weights are in the format of the setting
 * at the time they were packed: re-pack after a change (wtpse_hip/nn.py does).  Environment: WTPSE_X3_TERMS=1|2|3.  Other values only
 * query; returns the previous setting.  A recorded launch plan (wtpse_plan_*) remembers the setting it was recorded under and
 * refuses to replay under another one. */
int wtpse_x3_terms(int terms);
/* An "amax table" carries the largest magnitude of a gradient tensor from its producer to the x2h kernels that consume it: 1024
 * unsigneds (4 KB, 16-byte aligned) holding float bits of non-negative values in 64 shards, one per 64-byte line (waves / workgroups
 * fold their maximum into shard (index % 64) with one no-return atomic max; consumers take the maximum over the shards).  Producers with an `amax` argument
 * (wtpse_bn_bwd*, wtpse_upsample2x_bwd*) fill the table of the gradient they write when amax != NULL: the table must be ZERO on entry.
 * wtpse_amax zeroes and fills the table of an existing tensor (one extra pass: the slow way). */
int wtpse_amax(const float* x, long long n, unsigned* amax_table, void* stream);
int wtpse_pack_conv_weights_x3(const float* params, const int* desc, int n_desc, unsigned short* packed, void* stream);
int wtpse_conv_fwd_x3(const float* in0, int C0, const float* in1, int C1, const unsigned short* wpacked, const float* bias,
                      const float* pro0, const float* pro1, int pro_relu, float* out0, float* out1, int Csplit, float* stats,
                      int B, int H, int W, int Cout, int ksize, int relu_out, const float* mask_ref, const unsigned* in_amax,
                      const unsigned* in_amax1, unsigned* out_amax, void* stream);      /* in_amax / in_amax1 / out_amax: wtpse_x3_terms, wtpse_conv_fwd */

/* The 16-channel 3x3 layers (inc, DeepWT, the teacher's inc: algorithms.py:897-917,1091-1117,398-413) on v_mfma_f32_16x16x32_* with
 * register-resident weight fragments (csrc/conv.hip, MODE 3 / 4): Cout <= 16, C0 <= 16, one input.  wx16: the layer's fragments
 * (forward or data-gradient direction) from wtpse_pack_conv16_x3, 12808 unsigned shorts per direction (a 16-byte header with the
 * x2h weight scale, the x3 fragments, the x2h fragments), 16-byte aligned.  Arithmetic: wtpse_x3_terms() — x2h when it is 2, except
 * that a GRADIENT input (in_is_grad != 0) without its amax table (in_amax == NULL) runs in x3 (no scale is known for it and a pass
 * to find one costs more than this HBM-bound kernel would gain); in_is_grad == 0: a forward activation, scaled from in_amax (its
 * bound as loaded; NULL: 2^2).  out_amax: as wtpse_conv_fwd.
 * Options as wtpse_conv_fwd (bias, prologue, relu_out, stats [wtpse_conv_stats_blocks][Cout][2], mask_ref) plus gram_partial
 * (as wtpse_conv_fwd_gram; Cout == 16, relu_out == 0) and — with bn_mean — the BatchNorm-backward epilogue of wtpse_dgrad_bnb
 * over all output channels (mask_ref = that layer's raw conv output, stats = its partials). */
int wtpse_conv16_x3(const float* in0, int C0, const unsigned short* wx16, const float* bias, const float* pro0, int pro_relu,
                    float* out0, float* stats, float* gram_partial, const float* mask_ref, const float* bn_ss,
                    const float* bn_mean, int bn_relu, int B, int H, int W, int Cout, int relu_out, int in_is_grad,
                    const unsigned* in_amax, unsigned* out_amax, void* stream);
/* desc: n_desc x 8 ints {w_off, Cout, Cin, 9, fwd_off (-1: none), dgrad_off (-1: none), 0, 0}; w_off in floats, *_off in
 * unsigned shorts (multiples of 8). */
int wtpse_pack_conv16_x3(const float* params, const int* desc, int n_desc, unsigned short* packed, void* stream);

/* A data gradient (= wtpse_conv_fwd / wtpse_conv_fwd_x3 on dY with the `wd` / x3 data-gradient layout: C = the conv's output
 * channels, Cout = its input channels) that also performs the FIRST HALF of the BatchNorm backward of the conv + BatchNorm
 * (+ReLU) layer the gradient flows into (autograd of algorithms.py:883-889,904-917): output channels [bn_c0, bn_c1) — all of
 * them, or exactly one side of the Csplit split — are masked with the ReLU of that layer (bn_relu: [fmaf(bn_y, scale, shift)
 * > 0], bn_ss [Cbn][2], bn_y [B][Cbn][H][W] = its raw conv output, Cbn = bn_c1 - bn_c0) and the per-workgroup partials
 * (sum g, sum g * (bn_y - bn_mean)) are written to stats [wtpse_conv_stats_blocks | wtpse_conv_x3_stats_blocks][Cbn][2]:
 * wtpse_bn_bwd_from_stats finishes the BatchNorm backward without re-reading the two tensors for the reductions.
 * bn_c0, bn_c1 multiples of 16 (bn_c1 may equal Cout). */
int wtpse_dgrad_bnb(const float* dy, int C, const float* wpacked, float* out0, float* out1, int Csplit, const float* bn_y,
                    const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1, float* stats, int B, int H, int W,
                    int Cout, int ksize, void* stream);
int wtpse_dgrad_x3_bnb(const float* dy, int C, const unsigned short* wpacked, float* out0, float* out1, int Csplit,
                       const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1, float* stats,
                       int B, int H, int W, int Cout, int ksize, const unsigned* in_amax, void* stream);

/* A forward convolution in front of a train-mode BatchNorm (algorithms.py:883-889: conv -> bn) whose launch forms the
 * (sum, sum^2) partials of its output in `stats` AND finishes them — wtpse_bn_finalize's work, done by the workgroups that arrive
 * last (two levels of tickets, fixed fold order: bitwise reproducible; see wtpse_dgrad_bnb_coef below for partial2 / tickets,
 * sized with the same two queries).  layout: 0 = wtpse_conv_fwd (fp32 `wf`), 1 = wtpse_conv_fwd_x3, 2 = wtpse_conv16_x3's
 * fragments (one input).  No split, mask or ReLU output: the BatchNorm follows.  in_amax0 / in_amax1: the x2h input bounds of in0 /
 * in1 (layouts 1, 2; NULL: none); act_amax: NULL or the amax table (ZERO on entry) that receives the bound of the BatchNorm's
 * activated output, max_c |gamma_c| sqrt(B H W - 1) + |beta_c| (wtpse_x3_terms: the scale its x2h consumers load it with). */
int wtpse_conv_fwd_bnf(const float* in0, int C0, const float* in1, int C1, const void* wpacked, int layout, const float* bias,
                       const float* pro0, const float* pro1, int pro_relu, float* out0, float* stats, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, long long* num_batches, float momentum, float eps,
                       float* scale_shift, float* save_mean, float* save_invstd, double* partial2, unsigned* tickets, int B, int H,
                       int W, int Cout, int ksize, const unsigned* in_amax0, const unsigned* in_amax1, unsigned* act_amax, void* stream);

/* The same launches, which then ALSO finish the statistics: groups of 64 workgroups fold their partials as their last member
 * arrives, the last group of an output-channel block folds the group sums (fixed order: bitwise reproducible, nobody waits) and
 * writes coef [Cbn][3] = (k1, k2, k3) and dgamma / dbeta (+)= (accumulate) of the BatchNorm'd channels — what
 * wtpse_bn_bwd_from_stats does in its first launch.  wtpse_bn_bwd_apply_coef is then the whole rest of the BatchNorm backward.
 * (Launches of more than 8192 workgroups — WTPSE_TAIL_MAX_WGS; 2048 until round 6 — run that fold as a second launch inside the call instead: the
 * hand-off costs every workgroup ~2.5 us of lifetime, which beats a 6 us launch only where a CU sees few workgroups.)
 * layout: 0 = wtpse_dgrad_bnb (fp32 `wd`), 1 = wtpse_dgrad_x3_bnb, 2 = wtpse_conv16_x3's fragments (Csplit == Cout, all channels).
 * gamma / invstd: of the BatchNorm'd channels.  partial2: wtpse_bnb_tail_partial2(nblk, Cout) doubles of scratch; tickets:
 * wtpse_bnb_tail_tickets(nblk, Cout) unsigneds, ZERO on entry and zero again when the launch has finished (nblk = rows of stats);
 * two launches that may overlap must not share them. */
int wtpse_bnb_tail_partial2(int nblk, int Cout);
int wtpse_bnb_tail_tickets(int nblk, int Cout);
int wtpse_dgrad_bnb_coef(const float* dy, int C, const void* wpacked, int layout, float* out0, float* out1, int Csplit,
                         const float* bn_y, const float* bn_ss, const float* bn_mean, int bn_relu, int bn_c0, int bn_c1,
                         float* stats, const float* gamma, const float* invstd, float* coef, float* dgamma, float* dbeta,
                         int accumulate, double* partial2, unsigned* tickets, int B, int H, int W, int Cout, int ksize,
                         const unsigned* in_amax, void* stream);      /* in_amax: of dy, used by layout 1 (see wtpse_x3_terms) */

/* dW[Cout][C0+C1][k][k] (+)= sum dY * X, dbias (+)= sum dY (dbias/dbias_slab NULL: skip).  slab: [ksplit][Cout*Cin*k*k],
 * dbias_slab: [ksplit][Cout], ksplit = wtpse_wgrad_ksplit(...).  x inputs take the same prologue as the forward. */
int wtpse_conv_wgrad(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                     const float* pro1, int pro_relu, float* slab, float* dbias_slab, int ksplit, float* dw, float* dbias, int accumulate, int B, int H,
                     int W, int Cout, int ksize, void* stream);
int wtpse_wgrad_ksplit(int B, int H, int W, int Cin, int Cout);
/* The weight gradient in the x3 arithmetic of wtpse_conv_fwd_x3 (csrc/conv_x3.hip: dY and X kept pixel-major in LDS as bf16
 * triples, fragments fetched with the transposing LDS read).  3x3 convs with Cin, Cout multiples of 32 (C0 % 8 == 0 for a
 * concat): wtpse_wgrad_x3_supported().  No bias gradient (the conv+BatchNorm layers it serves have none).
 * slab: [ksplit][Cout*Cin*9], ksplit = wtpse_wgrad_x3_ksplit(...). */
int wtpse_wgrad_x3_supported(int Cin, int Cout, int ksize, int C0);
int wtpse_wgrad_x3_ksplit(int B, int H, int W, int Cin, int Cout);
int wtpse_conv_wgrad_x3(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                        const float* pro1, int pro_relu, float* slab, int ksplit, float* dw, int accumulate, int B, int H, int W,
                        int Cout, int ksize, void* stream);

/* The 3x3 weight gradient in the x3 arithmetic with register-resident operands (csrc/wgrad_r.hip: every lane loads its own
 * 8-pixel fragments from global memory, splits them into bf16 triples and feeds v_mfma_f32_16x16x32_bf16 from registers; the
 * horizontal taps are lane shifts, the vertical ones a 3-row ring of registers; no LDS in the main loop).  Maps whose width is a
 * multiple of 32, Cin / Cout multiples of 16 (C0 % 16 == 0 for a concat) — or exactly 16 wide with Cin / Cout multiples of 32 (two images
 * side by side per 32-pixel step; no bias gradient, not the _bn form) —: wtpse_wgrad_r_supported().  With bias gradient
 * (dbias / dbias_slab NULL: skip).  slab: [nslab][Cout*Cin*9], dbias_slab: [nslab][Cout], nslab = wtpse_wgrad_r_slabs(...).
 * Arithmetic by wtpse_x3_terms(); with 2 (x2h) dY is scaled from dy_amax (NULL: like a forward activation — and the 16 x 16-channel
 * blocks, HBM-bound, then stay on x3), X from x_amax0 / x_amax1 (the bounds of x0 / x1 as loaded; both NULL: 2^2); the _bn form below
 * stays on three bf16 terms.  With 1 (bf16 mode) only the blocks of
 * 32 channels on at least one side run with one term: the 16 x 16 blocks (which alone carry a bias gradient), the _bn form and
 * wtpse_conv_wgrad_x3 keep three — the mode is a mix of arithmetics by design (the 16-channel layers are not MFMA-bound). */
int wtpse_wgrad_r_supported(int Cin, int Cout, int ksize, int C0, int W);
int wtpse_wgrad_r_slabs(int B, int H, int W, int Cin, int Cout);
int wtpse_conv_wgrad_r(const float* dy, const float* x0, int C0, const float* x1, int C1, const float* pro0,
                       const float* pro1, int pro_relu, float* slab, float* dbias_slab, int nslab, float* dw, float* dbias,
                       int accumulate, int B, int H, int W, int Cout, const unsigned* dy_amax, const unsigned* x_amax0,
                       const unsigned* x_amax1, void* stream);    /* *_amax: wtpse_x3_terms */

/* wtpse_conv_wgrad_r with dY given as the un-applied second half of a BatchNorm backward (wtpse_bn_bwd_coef):
 * dY = k1[c] * g + k2[c] * bn_y + k3[c], bn_coef [Cout][3] = (k1, k2, k3): the BatchNorm-apply pass of the backward
 * (read g, read y, write dY) never runs, the kernel forms dY from the two tensors as it loads its A fragments. */
int wtpse_conv_wgrad_r_bn(const float* g, const float* bn_y, const float* bn_coef, const float* x0, int C0, const float* x1,
                          int C1, const float* pro0, const float* pro1, int pro_relu, float* slab, int nslab, float* dw,
                          int accumulate, int B, int H, int W, int Cout, void* stream);

/* ---- BatchNorm2d, eps 1e-5, momentum 0.1 (algorithms.py:862-864) ---------------------------------------------- */
/* train mode: fold the conv epilogue's partials -> scale_shift[C][2], save_mean/invstd[C]; update running stats
 * (unbiased variance) and num_batches_tracked (int64) when given.  act_amax: as wtpse_conv_fwd_bnf (NULL: none). */
int wtpse_bn_finalize(const float* stats_partial, int nblk, int C, long long count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, long long* num_batches, float momentum, float eps,
                      float* scale_shift, float* save_mean, float* save_invstd, unsigned* act_amax, void* stream);
/* act_amax (the WHOLE table is written: it need not be zero) = bound of |scale_shift[c][0] * y + scale_shift[c][1]| over a tensor y whose
 * amax table is raw_amax: max_c |scale| * amax + |shift| — the x2h input bound of an activation behind an eval-mode BatchNorm. */
int wtpse_act_bound(const float* scale_shift, int C, const unsigned* raw_amax, unsigned* act_amax, void* stream);
/* eval mode: scale_shift from the running statistics. */
int wtpse_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                         float eps, int C, float* scale_shift, void* stream);
/* z = act(y * scale + shift), materialised (scale_shift NULL: identity). */
int wtpse_affine_act(const float* y, const float* scale_shift, int relu, float* z, int B, int C, int HW, void* stream);
/* dz (grad wrt z = act(bn(y))) -> dgamma, dbeta, dy.  partial: [wtpse_bn_bwd_nsplit][C][2], coef: [C][3]. */
int wtpse_bn_bwd(const float* dz, const float* y, const float* scale_shift, int relu, const float* gamma,
                 const float* save_mean, const float* save_invstd, float* partial, float* coef, float* dgamma,
                 float* dbeta, int accumulate, float* dy, int B, int C, int HW, unsigned* amax, void* stream);
int wtpse_bn_bwd_nsplit(int B, int C, int HW);
/* Synchronised BatchNorm (data parallel, statistics over the global batch): the backward in two halves around the
 * caller's all-reduce of sums[C][2].  dgamma/dbeta receive this rank's share; dy uses the global sums and count. */
int wtpse_bn_bwd_reduce(const float* dz, const float* y, const float* scale_shift, int relu, const float* save_mean,
                        const float* save_invstd, float* partial, float* sums_local, int B, int C, int HW, void* stream);
int wtpse_bn_bwd_apply(const float* dz, const float* y, const float* scale_shift, int relu, const float* gamma,
                       const float* save_mean, const float* save_invstd, const float* sums_local, const float* sums_global,
                       long long count_global, float* coef, float* dgamma, float* dbeta, int accumulate, float* dy, int B,
                       int C, int HW, unsigned* amax, void* stream);
/* second half of a BatchNorm backward whose reductions came out of a data gradient's epilogue (wtpse_dgrad_bnb /
 * wtpse_dgrad_x3_bnb): g = the already-masked incoming gradient, stats_partial [nblk][C][2] = (sum g, sum g * (y - mean))
 * per workgroup; dgamma / dbeta (+)=, dy = k1 * g + k2 * y + k3.  coef: [C][3] scratch. */
int wtpse_bn_bwd_from_stats(const float* g, const float* y, const float* stats_partial, int nblk, const float* gamma,
                            const float* save_mean, const float* save_invstd, float* coef, float* dgamma, float* dbeta,
                            int accumulate, float* dy, int B, int C, int HW, unsigned* amax, void* stream);

/* partials [nblk][C][2] = (sum g, sum g (y - mean)) -> coef [C][3], dgamma / dbeta (+)=: the first launch of
 * wtpse_bn_bwd_from_stats on its own. */
int wtpse_bn_bwd_finalize_coef(const float* stats_partial, int nblk, int C, long long count, const float* gamma,
                               const float* save_mean, const float* save_invstd, float* coef, float* dgamma, float* dbeta,
                               int accumulate, void* stream);
/* dy = k1 * g + k2 * y + k3 with coef [C][3] from wtpse_dgrad_bnb_coef. */
int wtpse_bn_bwd_apply_coef(const float* g, const float* y, const float* coef, float* dy, int B, int C, int HW, unsigned* amax, void* stream);

/* ---- WT (whitening) loss: compute_whitening_loss + compute_MMD (algorithms.py:1277-1309,59-121;
 *      shape_networks.py:561-594,240-309) ------------------------------------------------------------------------ */
/* z [B,16,HW] -> gram[B][256] (incl. eps*I), v[B][120], offdiag[B], diag[B], rowval[D*n + 1] (fp64; the extra element is
 * an 8-byte ticket word of the launch that merges the MMD rows with the final sums), dmmd_dv[D*n][120],
 * losses[3] = {ins_offdiag, ins_diag, domain}.  partial: [B * wtpse_wt_split(B,HW,&chunk)][256]. */
int wtpse_wt_loss_fwd(const float* z, int B, int C, int HW, float eps, float margin, int domain_num, int per_domain,
                      float* partial, float* gram, float* v, float* offdiag, float* diag, double* rowval,
                      float* dmmd_dv, float* losses, void* stream);
int wtpse_wt_split(int B, int HW, int* chunk_out);
/* The same loss from partial Grams [B][S][256] that a conv epilogue already produced (wtpse_conv_fwd_gram): z is not read. */
int wtpse_wt_loss_fwd_partials(const float* partial, int S, int B, int HW, float eps, float margin, int domain_num,
                               int per_domain, float* gram, float* v, float* offdiag, float* diag, double* rowval,
                               float* dmmd_dv, float* losses, void* stream);
/* The two halves of wtpse_wt_loss_fwd (data parallel: all-gather of v and wtpse_mmd_fwd on the global rows in between).
 * wtpse_wt_final: losses[0..1] = this rank's share of the instance means (divided by Bnorm), losses[2] = sum(rowval). */
int wtpse_wt_gram_fwd(const float* z, int B, int C, int HW, float eps, float* partial, float* gram, float* v, float* offdiag,
                      float* diag, void* stream);
int wtpse_wt_final(const float* offdiag, const float* diag, int B, int Bnorm, float margin, const double* rowval, int R,
                   float* losses, void* stream);
/* dz (+)= d(w_off*g_off*ins_off + w_diag*g_diag*ins_diag + w_dom*g_dom*dom)/dz.  g_*: device scalars (NULL = 1).
 * Mws: [B][256] scratch.  accumulate & 1: add to dz; accumulate & 2 (with & 1): the incoming dz is a gradient wrt relu(z) and is
 * masked with [z > 0] first (the teacher reads relu(z2): algorithms.py:1066) — z is in registers here anyway. */
int wtpse_wt_loss_bwd(const float* z, int B, int C, int HW, float margin, int domain_num, int per_domain,
                      const float* gram, const float* offdiag, const float* diag, const float* dmmd_dv,
                      const float* g_off, const float* g_diag, const float* g_dom, float w_off, float w_diag, float w_dom,
                      float* Mws, float* dz, int accumulate, void* stream);
/* Fold per-map losses [nmaps][3] into out[4] = (ins_total, ins_off, ins_diag, dom) with the reference's quirks:
 * mode 0 = WT_PSE.update (algorithms.py:1259-1267), mode 1 = student incl. accumulator overwrite (shape_networks.py:546-548). */
int wtpse_wt_combine(const float* losses, int nmaps, float den, int mode, float* out, void* stream);
/* MMD alone on v [D*n][120] (data-parallel: after the all-gather of v). sum(rowval) is the loss. */
int wtpse_mmd_fwd(const float* v, int domain_num, int per_domain, double* rowval, float* dmmd_dv, void* stream);

/* ---- pooling / upsampling (algorithms.py:890,901,929,949) ------------------------------------------------------- */
int wtpse_maxpool2_fwd(const float* x, const float* pro, int relu, float* out, int B, int C, int H, int W, void* stream);
/* accumulate & 1: dx += ; accumulate & 2: the result (after the add) is masked with [act(x) > 0], act = the prologue as loaded —
 * the backward of the ReLU that produced x (algorithms.py:1068-1069: fusion conv -> ReLU -> U-Net), fused. */
int wtpse_maxpool2_bwd(const float* x, const float* pro, int relu, const float* dout, float* dx, int accumulate, int B,
                       int C, int H, int W, void* stream);
/* wtpse_maxpool2_bwd(accumulate | 2) where x is the raw output of a conv + BatchNorm + ReLU layer (pro = its scale/shift): the
 * result is the masked gradient wrt that layer's activated output, and the launch also forms the two reductions of its BatchNorm
 * backward: stats [wtpse_maxpool2_bwd_stats_blocks(B,H,W)][C][2] = (sum g, sum g (y - mean)) partials, the layout
 * wtpse_bn_bwd_from_stats folds.  H even, W % 4 == 0, 16-byte aligned tensors, B*C < 32768. */
int wtpse_maxpool2_bwd_stats_blocks(int B, int H, int W);
int wtpse_maxpool2_bwd_bnb(const float* x, const float* pro, int relu, const float* dout, float* dx, int accumulate,
                           const float* mean, float* stats, int B, int C, int H, int W, void* stream);
/* bilinear x2, align_corners=False; H, W are the INPUT sizes. */
int wtpse_upsample2x_fwd(const float* x, const float* pro, int relu, float* out, int B, int C, int H, int W, void* stream);
int wtpse_upsample2x_bwd(const float* dout, float* dx, int accumulate, int B, int C, int H, int W, unsigned* amax, void* stream);
/* ... of a gradient that is the second half of a BatchNorm backward and never written out: dout = k1[c] * g + k2[c] * bn_y + k3[c],
 * bn_coef [C][3] as wtpse_dgrad_bnb_coef leaves it (the expression of wtpse_bn_bwd_apply_coef, same bits); g, bn_y [B][C][2H][2W].
 * W % 4 == 0, 16-byte aligned tensors. */
int wtpse_upsample2x_bwd_bn(const float* g, const float* bn_y, const float* bn_coef, float* dx, int B, int C, int H, int W,
                            unsigned* amax, void* stream);
/* The same with the train-mode BatchNorm statistics of the OUTPUT: stats [wtpse_upsample2x_stats_blocks(B,H,W)][C][2]
 * per-workgroup (sum, sum of squares), the layout wtpse_bn_finalize takes.  Used where the 1x1 conv of a ConvU block
 * (algorithms.py:949-951: upsample -> conv2 -> bn2) runs in front of the upsampling instead (the two commute). W even. */
int wtpse_upsample2x_fwd_stats(const float* x, float* out, float* stats, int B, int C, int H, int W, void* stream);
int wtpse_upsample2x_stats_blocks(int B, int H, int W);

/* F.interpolate(size=(Ho,Wo), mode="bilinear"), align_corners=False: validation resize of the logits (Trainer.py:206-209). */
int wtpse_resize_bilinear(const float* x, float* out, int B, int C, int H, int W, int Ho, int Wo, void* stream);

/* ---- shape attention + fusion (algorithms.py:1126-1129,1243-1248,1342-1344) ------------------------------------- */
/* att = sigmoid(w*z + b), fuse = coef*emb + att*emb, mask = att > 0.75.  wb: device {w, b}.  att/att_pre/mask optional. */
int wtpse_attn_fuse_fwd(const float* z, const float* wb, const float* emb, float coef, float* att, float* att_pre,
                        float* mask, float* fuse, int B, int CE, int HW, void* stream);
/* partial: [2*ceil(B*HW/256)]; d_wb[2] (+)= (dw, db); dz optional. */
int wtpse_attn_fuse_bwd(const float* dfuse, const float* z, const float* emb, const float* att, const float* wb, float coef,
                        float* demb, float* dz, float* partial, float* d_wb, int accumulate, int B, int CE, int HW,
                        void* stream);

/* ---- sampling (algorithms.py:1068-1075; shape_networks.py:490-510) ----------------------------------------------- */
int wtpse_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, long long n, void* stream);
int wtpse_reparam_bwd(const float* dz, const float* logvar, const float* eps, float* dlogvar, long long n, void* stream);
int wtpse_exp_half(const float* logvar, float* std_, long long n, void* stream);
int wtpse_reparam_student(const float* mu, const float* std_, const float* eps, float* z, long long n, void* stream);
/* if any element is NaN: nan_to_num the whole tensor (shape_networks.py:490-492: `if torch.isnan(mu).any(): mu = nan_to_num(mu)`).
 * flag: one device int (scratch).  No host sync. */
int wtpse_nan_scrub(float* x, long long n, int* flag, void* stream);
/* Philox4x32-10 + Box-Muller; element i depends only on (seed, position + i): position = offset + *offset_dev (offset_dev
 * NULL: 0) = global element index, a multiple of 4.  Keeping the running position in device memory lets a captured
 * launch (hipGraph) draw fresh numbers on every replay. */
int wtpse_randn(float* out, long long n, unsigned long long seed, unsigned long long offset,
                const unsigned long long* offset_dev, void* stream);
/* *counter += inc on the stream (int32 counter, or uint64 when is64): step / stream-position counters of captured launches. */
int wtpse_counter_add(void* counter, long long inc, int is64, void* stream);

/* ---- caller-side losses and optimiser (Trainer.py:19,787,842-871; shape_networks.py:596-597; train.py:120-138) --- */
/* `partial` scratch: wtpse_reduce_blocks(n) floats (x2 for wtpse_pos_weight).  g: device scalar upstream gradient or NULL. */
int wtpse_reduce_blocks(long long n);
int wtpse_bce_sigmoid_fwd(const float* x, const float* t, long long n, float* partial, float* loss, void* stream);
int wtpse_bce_sigmoid_bwd(const float* x, const float* t, const float* g, float w, long long n, float* dx, void* stream);
int wtpse_pos_weight(const float* mask, const float* t, long long n, float* partial, float* sums, float* pw, void* stream);
int wtpse_pos_weight_from_sums(const float* sums, float* pw, void* stream);
int wtpse_bce_logits_pw_fwd(const float* x, const float* mask, const float* t, const float* pw, long long n, float* partial,
                            float* loss, void* stream);
int wtpse_bce_logits_pw_bwd(const float* x, const float* mask, const float* t, const float* pw, const float* g, float w,
                            long long n, float* dx, void* stream);
int wtpse_mse_fwd(const float* a, const float* b, long long n, float* partial, float* loss, void* stream);
int wtpse_mse_bwd(const float* a, const float* b, const float* g, float w, long long n, float* da, void* stream);
int wtpse_roi(const float* image, const float* logit, float* roi, float* od_pred, int B, int C, int HW, void* stream);
/* torch.optim.Adam (no weight decay / amsgrad), step number t = step + *step_dev (step_dev NULL: t = step; otherwise a
 * device int holding the number of completed steps, so that a captured launch stays valid when replayed). */
int wtpse_adam(float* p, const float* g, float* m, float* v, long long n, double lr, double beta1, double beta2, double eps,
               int step, const int* step_dev, void* stream);

/* ---- fused 1x1 heads (csrc/head.hip) ------------------------------------------------------------------------------ */
/* The heads 32 -> 32 (ReLU) -> 8 [-> (ReLU) -> nc] as one kernel per direction: reference algorithms.py:1006-1012
 * (mu_prior / logvar_prior, three layers, nc <= 4) and :1199-1200 (the segmentation net's `mu`, two layers: w3 = b3 = y =
 * NULL and the 8-channel result is h2).  Weights are the plain OIHW parameters ([32][32], [8][32], [nc][8]), HW % 32 == 0.
 * x takes the conv loaders' prologue (pro: [32][2] scale/shift or NULL, pro_relu).  h1 ([B][32][HW], post-ReLU) and h2
 * ([B][8][HW], post-ReLU for three layers) are the tapes: pass NULL to skip storing them (h2 is mandatory for a two-layer
 * head: it is the output).
 * Arithmetic (round 6): under x2h (wtpse_x3_terms() == 2, the default) both layers run on the fp16 matrix cores at fp32 accuracy
 * exactly as the convolutions do — every operand = two fp16 terms of a power-of-two multiple of the value, three products, fp32
 * accumulation (12 v_mfma_f32_32x32x16_f16 per 32-pixel block where the fp32-input form needs 32 v_mfma_f32_32x32x2_f32: the
 * kernels were bound by those, not by their bytes).  x_amax: the amax table BOUNDING THE ACTIVATED INPUT (wtpse_amax /
 * wtpse_act_bound / a producer's table), or NULL: the fixed forward scale 2^2, i.e. |act(x)| < 2^13 or the outputs are NaN; W1 / W2
 * are scaled by their own largest magnitude, h1 by max_m sum_k |W1[m][k]| * bound(x) + max |b1|.  In the other modes (x3, bf16)
 * the fp32-input kernels of rounds 2-5 run and x_amax is ignored. */
int wtpse_head_fwd(const float* x, const float* pro, int pro_relu, const float* w1, const float* b1, const float* w2,
                   const float* b2, const float* w3, const float* b3, int nc, float* h1, float* h2, float* y,
                   const unsigned* x_amax, int B, int HW, void* stream);
/* dy: [B][nc][HW] (three layers) or [B][8][HW] (two layers: gradient of the h2 output).  dx: [B][32][HW] gradient wrt the
 * activated input.  dparams: [32*32 + 32 + 8*32 + 8 (+ 8*nc + nc)] = (dW1, db1, dW2, db2[, dW3, db3]) contiguous, which is
 * the order the head's parameters have in the flat gradient buffer; written, or added to when accumulate != 0.
 * slab: scratch of wtpse_head_slabs(B, HW) * that many floats.
 * x2h: h1 is NOT read (may be NULL; the forward need not store it) — layer 1 is formed again from the x the kernel reads anyway, by
 * the forward kernel's instruction sequence on its operands (the same bits, so the forward's ReLU mask): b1 and dy_amax (the amax
 * table of dy) are required, x_amax as in wtpse_head_fwd (the SAME table, or the masks may differ in the last bit of h1).  The
 * gradients' scales follow from bound(dy) through the weights' absolute row sums.  Other modes: h1 required, b1 / tables ignored. */
int wtpse_head_bwd(const float* dy, const float* x, const float* pro, int pro_relu, const float* h1, const float* h2,
                   const float* w1, const float* b1, const float* w2, const float* w3, int nc, float* dx, float* slab,
                   float* dparams, int accumulate, const unsigned* x_amax, const unsigned* dy_amax, int B, int HW, void* stream);
int wtpse_head_slabs(int B, int HW);

/* ---- device-side training input pipeline (csrc/pipeline.hip) ---------------------------------- */
/* One pass of Pillow's 8-bit separable resampling (the engine behind the reference's Image.resize calls,
 * custom_transforms.py:342-346,375-391).  in [N][Hin][Win][C] uint8; vertical = 0: out [N][Hin][L][C], vertical = 1:
 * out [N][L][Win][C].  bounds [T][L][2] = (first source index, tap count) and kk [T][L][ksize] = 22-bit fixed-point
 * coefficients as Pillow's precompute_coeffs / normalize_coeffs_8bpc produce them; sample n reads table tab[n]
 * (tab NULL: table 0 for every sample). */
int wtpse_resample_u8(const unsigned char* in, unsigned char* out, const int* bounds, const int* kk, const int* tab, int ksize,
                      int N, int Hin, int Win, int C, int L, int vertical, void* stream);
/* Normalize_tf + ToTensor (custom_transforms.py:455-499,581-599) on img [N][S][S][3] uint8 and the resized disc mask od
 * [N][S][S] uint8, read through per-sample index tables xidx / yidx [N][S] (NEAREST resize + crop of the mask):
 * image [N][3][S][S] = img / 127.5 - 1, od_out [N][1][S][S] = (mask <= 200), oc_out = (mask <= 50). */
int wtpse_input_finish(const unsigned char* img, const unsigned char* od, const int* xidx, const int* yidx, float* image,
                       float* od_out, float* oc_out, int N, int S, void* stream);

/* ---- small utilities ------------------------------------------------------------------------------------------- */
int wtpse_relu_mask(const float* dz, const float* ref, float* dy, int accumulate, long long n, void* stream);
int wtpse_axpy(float* dst, const float* src, float alpha, long long n, void* stream);
int wtpse_reduce_rows(const float* partial, int rows, int cols, float* out, int accumulate, float scale, void* stream);
int wtpse_zero(void* p, long long nbytes, void* stream);
/* dst[0:n] = src[0:n] (n % 4 == 0, 16-byte aligned) with 16 or 4 bytes per lane and access: measurement infrastructure only — the
 * known-byte-count streams the rocprofv3 FETCH_SIZE / WRITE_SIZE factors are calibrated on (tools/pmc_traffic.py, bench.py --kernels-only). */
int wtpse_copy_probe(const float* src, float* dst, long long n, int bytes_per_lane, void* stream);

/* ---- standalone 2-D discrete wavelet transform (csrc/dwt.hip) — NOT part of WT-PSE ---------------------------------------
 * The reference has no wavelet transform (its "WT" is the whitening transform); these two entry points exist only as the
 * HBM-bandwidth micro-benchmark BASELINE.json's wording names and are never called by the training path.
 * Specification (self-defined, parity unpinned): oracle/dwt_cpu.py — separable, orthonormal, periodic extension, lifting,
 * Mallat layout.  x / coef: [planes][H][W] fp32, distinct buffers; wavelet 0 = Haar, 1 = Daubechies 4-tap ("db2");
 * H, W divisible by 2^levels; tmp: 2 * planes * (H/2) * (W/2) floats. */
int wtpse_dwt2_fwd(const float* x, float* coef, float* tmp, int planes, int H, int W, int wavelet, int levels, void* stream);
int wtpse_dwt2_inv(const float* coef, float* x, float* tmp, int planes, int H, int W, int wavelet, int levels, void* stream);

/* ---- launch plans (csrc/plan.hip): a recorded sequence of the calls above, replayed with one host call ------------------
 * wtpse_plan_add_call: `fn` indexes the entry points that take a stream (wtpse_plan_fn_name(fn), 0 <= fn < wtpse_plan_fn_count());
 * `args`: its arguments without the trailing stream, one 8-byte slot each (pointer / long long / unsigned long long / double).
 * wtpse_plan_add_wait: stream `waiter` waits for everything issued so far on stream `waited`. */
int wtpse_plan_fn_count(void);
const char* wtpse_plan_fn_name(int id);
void* wtpse_plan_create(void);
int wtpse_plan_destroy(void* plan);
int wtpse_plan_size(void* plan);
int wtpse_plan_add_call(void* plan, int fn, const void* args, int nargs, void* stream);
int wtpse_plan_add_wait(void* plan, void* waiter, void* waited);
int wtpse_plan_replay(void* plan);      /* -2 (WTPSE_ESTATE): wtpse_tuning_state() differs from the recording's */
/* The run-time switches that decide tilings and the packed-weight format (wtpse_x3_terms | wtpse_x3r_enable << 4 | wtpse_x3_xcd << 8 |
 * wtpse_x3_small_wide << 12). */
int wtpse_tuning_state(void);

/* Fingerprint (hex) of the sources and of this header the library was compiled from; the binding refuses a library
 * whose fingerprint differs from the tree's (a stale .so after a signature change would otherwise go unnoticed). */
const char* wtpse_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif
