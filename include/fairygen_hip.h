/*
 * fairygen_hip.h — C ABI of libfairygen_hip.so, the MI355X (gfx950) kernels behind FairyGen's animation
 * hot path (diffsynth WanVideoPipeline -> Wan2.2-TI2V-5B denoise loop -> Wan2.2 VAE decode).
 *
 * The reference is pure Python/PyTorch and has no FFI; each entry point below replaces the ATen op
 * sequence of one reference call site (cited per function, paths relative to
 * /root/reference/animation/).  A host binds them with ctypes:
 * tensor.data_ptr() for buffers, torch.cuda.current_stream().cuda_stream for `stream`.
 *
 * Contract (all functions):
 *   - return 0 on success, a negative FG_E* code on error; fg_last_error() gives a thread-local message;
 *   - no allocation, no ownership transfer, no host synchronisation: the caller passes every buffer,
 *     work is enqueued asynchronously on `stream` (a hipStream_t; NULL = the default stream);
 *   - re-entrant across streams and devices, no mutable global state;
 *   - all tensors are contiguous row-major device buffers unless a leading dimension (`ld*`, in
 *     ELEMENTS) is given; bf16 = IEEE bfloat16 bits; pointers must be 16-byte aligned and the channel
 *     counts multiples of 8 (checked, FG_EINVAL);
 *   - token tensors are "b s (n d)" exactly like the reference's AttentionModule
 *     (models/wan_video_dit.py:113-120); VAE activations are channels-last (T,H,W,C) inside the
 *     decoder, NCTHW only at fg_vae_* boundary kernels.
 */
#ifndef FAIRYGEN_HIP_H
#define FAIRYGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FG_OK        0
#define FG_EINVAL   (-1)   /* bad shape / alignment / unsupported size */
#define FG_ELAUNCH  (-2)   /* hipLaunch failed (message has the HIP error string) */

typedef void* fg_stream_t;   /* hipStream_t */

int         fg_version(void);            /* ABI version, currently 3 */
const char* fg_last_error(void);         /* thread-local, valid until the next failing call */

/* ------------------------------------------------------------------ DiT token-side kernels (HBM-bound)
 * "Modulation rows": AdaLN shift/scale/gate vectors live in a small (mod_rows, *, C) table; row r of a
 * vector starts at ptr + r*mod_ld elements.  mod_rows==1: one row for all tokens (T2V mode);
 * mod_rows==2: tokens [0,first_rows) use row 0 and the rest row 1 (TI2V mode: first latent frame is at
 * t=0, pipelines/wan_video.py:1218-1228); mod_rows==rows: one row per token (the reference's literal
 * (1,N,6,C) layout). */

/* out = bf16(bf16(bf16(LN(x)) * bf16(1+scale)) + shift): nn.LayerNorm(elementwise_affine=False) followed by
 * modulate(), models/wan_video_dit.py:63-64,205-206,224,227 and Head.forward :261-268. */
int fg_ln_modulate_bf16(const void* x, const void* shift, const void* scale, void* out,
                        int64_t rows, int C, float eps,
                        int64_t mod_rows, int64_t first_rows, int64_t mod_ld, fg_stream_t stream);

/* out = LN(x)*w + b (norm3, models/wan_video_dit.py:207,226). */
int fg_ln_affine_bf16(const void* x, const void* w, const void* b, void* out,
                      int64_t rows, int C, float eps, fg_stream_t stream);

/* out = x + gate*y (GateModule, models/wan_video_dit.py:188-193,225,228) or x + y when gate==NULL (:226).
 * out may alias x. */
int fg_gate_residual_bf16(const void* x, const void* y, const void* gate, void* out,
                          int64_t rows, int C,
                          int64_t mod_rows, int64_t first_rows, int64_t mod_ld, fg_stream_t stream);

/* Fused residual + next norm: x_out = x + gate*y (gate may be NULL); then
 *   mode 0: norm_out = LN(x_out) modulated by (shift, scale) as fg_ln_modulate_bf16,
 *   mode 1: norm_out = LN(x_out)*w + b (p0 = w, p1 = b; mod_* ignored for the norm).
 * One pass over HBM instead of two (models/wan_video_dit.py:225-227, and :228 -> next block's :224).
 * x_out may alias x. */
int fg_residual_ln_bf16(const void* x, const void* y, const void* gate, void* x_out,
                        const void* p0, const void* p1, void* norm_out, int mode,
                        int64_t rows, int C, float eps,
                        int64_t mod_rows, int64_t first_rows, int64_t mod_ld, fg_stream_t stream);

/* fg_ln_modulate_bf16 / fg_residual_ln_bf16 for a norm whose ONLY consumer is an fp8 Linear (the fp8 mode of the DiT blocks:
 * AutoWrappedLinear.fp8_linear, core/vram/layers.py:331-342, after models/wan_video_dit.py:224-228): instead of the bf16 row, the
 * normalised row leaves as e4m3 bytes out_fp8 / norm_fp8 (rows, C) and one fp32 scale per row — bit for bit what
 * fg_fp8_quant_rows_bf16 (act 0) makes of the bf16 row, without writing and re-reading it.  fp8_max: 448 for e4m3fn. */
int fg_ln_modulate_fp8_bf16(const void* x, const void* shift, const void* scale, void* out_fp8, float* out_scale,
                            int64_t rows, int C, float eps, int64_t mod_rows, int64_t first_rows, int64_t mod_ld,
                            float fp8_max, fg_stream_t stream);
int fg_residual_ln_fp8_bf16(const void* x, const void* y, const void* gate, void* x_out, const void* p0, const void* p1,
                            void* norm_fp8, float* norm_scale, int mode, int64_t rows, int C, float eps,
                            int64_t mod_rows, int64_t first_rows, int64_t mod_ld, float fp8_max, fg_stream_t stream);

/* nn.Linear of the DiT blocks (models/wan_video_dit.py:130-133,156-159,208-209): c[M,N] = a[M,K] w[N,K]^T + bias[N], bf16 in and
 * out, fp32 accumulation, bias added to the accumulator before the bf16 rounding — alone or with the op that follows it in the block
 * folded into the store (:188-193 GateModule, :208 nn.GELU(approximate='tanh'), :225-228):
 *   mode 0: c = bf16(acc + bias)                                   (nn.Linear)
 *   mode 2: c = bf16(c + bf16(gate * bf16(acc + bias)))            (x = x + gate * Linear(.), x read in place from c)
 *   mode 3: c = bf16(c + bf16(acc + bias))                         (x = x + Linear(.))
 *   mode 4: c = bf16(gelu_tanh(bf16(acc + bias)))                  (ffn.0 + GELU)
 * with the bf16 rounding points of the reference's separate ops.  a has leading dimension lda, c ldc (elements); w is the row-major
 * (out_features, in_features) weight.  gate: a table of gate_rows (1 or 2) rows of N values, row stride gate_ld elements; with 2 rows,
 * output rows < first_rows use row 0 (the first latent frame's t = 0 modulation), the others row 1.  N %% 256 == 0, K %% 128 == 0.
 * Persistent kernel: one workgroup per CU of the device takes 256 x 256 output tiles from a per-XCD cursor (csrc/gen_gemm_p.py).  The tiles left
 * over after the last whole round of the CUs are finished as pieces: with `workspace` (fg_gemm_workspace_bytes(M, N, K) bytes of
 * device memory, 16-byte aligned; may be NULL) and K >= 6144 or at most 8 rounds of tiles per CU cut along K into fp32 partial sums that a second small kernel adds in k
 * order before the same epilogue — those elements' fp32 summation is then grouped per piece (fixed for a given shape and device; every
 * other element is one full-K accumulation in k order); otherwise cut into 64-column pieces, every element accumulated in k order. */
int64_t fg_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);
int fg_gemm_epilogue_bf16(const void* a, int64_t lda, const void* w, const void* bias, void* c, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, int mode, const void* gate, int64_t gate_rows, int64_t gate_ld,
                          int64_t first_rows, void* workspace, fg_stream_t stream);

/* Diagnostics for the persistent GEMM's unit scheduler: launch only `workgroups` workgroups (a multiple of the XCD count, at most one
 * per CU; 0 = one per CU again).  The units of a launch are fixed by the shape; workgroups take them from per-XCD cursors, so the result
 * is bit-identical with any number of workgroups — as when other kernels (RCCL) hold some CUs.  The library keeps 256 bytes of device
 * memory per (device, stream) it has launched a GEMM on (the cursors; cleared by the last workgroup of every launch). */
int fg_gemm_debug_grid(int workgroups);

/* The matmul of AutoWrappedLinear.fp8_linear (core/vram/layers.py:343-357: torch._scaled_mm(x_fp8, w_fp8.T, scale_a (rows, 1),
 * scale_b = ones (1, out), bias, out_dtype = bf16)) on the same persistent kernel with e4m3 operands (OCP float8_e4m3fn,
 * v_mfma_f32_32x32x64_f8f6f4): y = bf16((acc * scale_a[row]) + bias[col]) — one fp32 rounding per operation, as the library op —
 * followed by the epilogue `mode` of fg_gemm_epilogue_bf16 (0, 2, 3, 4) on y.  a_fp8: (M, K) bytes with leading dimension lda,
 * w_fp8: (N, K) row-major, scale_a: M fp32 values (fg_fp8_quant_rows_bf16 / the fp8-output norm kernels produce a_fp8 and scale_a).
 * N %% 256 == 0, K %% 256 == 0, lda %% 16 == 0; workspace as above (taken for K >= 12288 or at most 8 rounds of tiles per CU). */
int fg_gemm_fp8_bf16(const void* a_fp8, int64_t lda, const float* scale_a, const void* w_fp8, const void* bias, void* c, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int mode, const void* gate, int64_t gate_rows, int64_t gate_ld,
                     int64_t first_rows, void* workspace, fg_stream_t stream);

/* RMSNorm over the full row (all heads), * weight, then optional 3-D RoPE on adjacent pairs:
 * RMSNorm.forward models/wan_video_dit.py:99-110 + rope_apply :91-96 (SelfAttention.forward :140-144,
 * CrossAttention.forward :176-177 with cos==sin==NULL).  x has leading dimension ldx (so q/k slices of a
 * fused QKV projection work), out is (rows, C) contiguous.
 * table_f32 == 0: cos_tab / sin_tab are fp64 (rows, head_dim/2) tables, the real / imag parts of the reference's
 *   complex128 table (pipelines/wan_video.py:1271-1275), and the rotation is done in fp64 exactly like rope_apply.
 * table_f32 != 0 (the default of the pipeline): cos_tab is ONE interleaved fp32 table (rows, head_dim/2, {cos, sin}) —
 *   the fp64 table rounded once — sin_tab is NULL, and the rotation is two fp32 FMAs per output: the fp64 rotation
 *   costs 43 us of VALU per call on top of the 77 us HBM time at N = 27 280 (2.8 -> 4.3 TB/s); the bf16 output
 *   differs from the fp64 mode in < 0.2 % of the elements, by one bf16 ulp (tests/test_hip_kernels.py). */
int fg_rmsnorm_rope_bf16(const void* x, int64_t ldx, const void* weight,
                         const void* cos_tab, const void* sin_tab, int table_f32, void* out,
                         int64_t rows, int C, int num_heads, float eps, fg_stream_t stream);

/* Same arithmetic, head-group-major output for the sequence-parallel Ulysses exchange (the reference's
 * xfuser path re-lays q/k out inside xFuserLongContextAttention, utils/xfuser/xdit_context_parallel.py:125-146):
 * column block g = col / group_cols of row r is written to out + g*out_group_stride + r*out_ld + col % group_cols,
 * i.e. straight into the all-to-all send buffer (block g = the heads of rank g). */
int fg_rmsnorm_rope_grouped_bf16(const void* x, int64_t ldx, const void* weight,
                                 const void* cos_tab, const void* sin_tab, int table_f32, void* out,
                                 int64_t rows, int C, int num_heads, float eps,
                                 int group_cols, int64_t out_group_stride, int64_t out_ld, fg_stream_t stream);

/* dst[g*dst_group_stride + r*dst_ld + c] = src[g*src_group_stride + r*src_ld + c] for g < groups, r < rows, c < cols
 * (strides in elements): the v columns into the Ulysses send buffer, and the received head-group blocks back to
 * "b s (n d)" rows before the o projection (xdit_context_parallel.py:139-145, the reference's rearranges). */
int fg_copy_groups_bf16(const void* src, int64_t src_group_stride, int64_t src_ld,
                        void* dst, int64_t dst_group_stride, int64_t dst_ld,
                        int groups, int64_t rows, int cols, fg_stream_t stream);

/* Activation side of the reference's fp8 Linear, AutoWrappedLinear.fp8_linear (core/vram/layers.py:321-357): per row of
 * x (rows, C; leading dimension ldx) scale[r] = max(bf16(max|x_r| / fp8_max), 1) and out_fp8[r] = e4m3fn(x_r / (scale[r] + 1e-8))
 * (OCP e4m3, torch.float8_e4m3fn; round to nearest even) — the arguments of the torch._scaled_mm call the reference makes
 * (row-wise scale_a, unit scale_b, bf16 bias).  act 1 applies GELU(tanh) first (ffn.2's input = nn.GELU(ffn.0(x)), rounded
 * to bf16 like the module output; optionally also written to act_out (rows, C)), act 0 quantises x as is.  fp8_max = 448. */
int fg_fp8_quant_rows_bf16(const void* x, int64_t ldx, void* out_fp8, float* scale, void* act_out,
                           int64_t rows, int C, int act, float fp8_max, fg_stream_t stream);

/* Elementwise activation, out may alias x.  kind 0: SiLU (time_embedding / time_projection,
 * models/wan_video_dit.py:312-318); kind 1: GELU(tanh) (ffn / text_embedding, :208-209,307-311). */
int fg_act_bf16(const void* x, void* out, int64_t n, int kind, fg_stream_t stream);

/* Non-causal softmax attention, head_dim 128: flash_attention()/AttentionModule,
 * models/wan_video_dit.py:27-60,113-120 (SDPA semantics: softmax(q k^T * scale) v, fp32 accumulate).
 * q:(B,Nq,H*128) k,v:(B,Nkv,H*128) with leading dimensions ldq/ldk/ldv (elements between token rows) and
 * batch strides Nq*ldq etc.; out (B,Nq,H*128) contiguous.  bf16 MFMA, fp32 online softmax.
 * workspace (optional, may be NULL/0): fg_attn_workspace_bytes(B,Nq,Nkv,H) bytes of scratch let the launcher cut
 * the tail q-blocks (or, for short query ranges such as a 1/8 token shard, every q-block) into KV ranges that are
 * merged by a second small kernel, so that the workgroup count fills the 256 CUs evenly; results are the same up
 * to fp32 summation order.  K/V of one batch element must span < 4 GiB.
 * scale: any positive value.  Self-attention shapes (Nkv > 1024) run a hand-scheduled 4-wave kernel in one of two forms with the same
 * result contract: when scale * log2(e) is a power of two (to 1e-6: pass 2^e / log2(e)) Q^T is pre-multiplied by it — exact in bf16 — and
 * the exponent needs no per-score multiply; for any other scale (e.g. 1/sqrt(128)) the scores of the exact operands are scaled in fp32.
 * A caller that wants the faster form for 1/sqrt(d) folds the factor 2^-e / sqrt(d) * log2(e) into q itself (wan_video_dit.py does,
 * through q's RoPE table). */
int64_t fg_attn_workspace_bytes(int B, int64_t Nq, int64_t Nkv, int H);
/* Diagnostic: the decomposition the launcher picks for this shape and workspace size: the last R q-blocks of every
 * (batch, head) are cut into S KV ranges (R == 0: no split). */
int fg_attn_split_choice(int B, int64_t Nq, int64_t Nkv, int H, int64_t workspace_bytes, int* R, int* S);
int fg_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                     void* out, int B, int64_t Nq, int64_t Nkv, int H, int D, float scale,
                     void* workspace, int64_t workspace_bytes, fg_stream_t stream);

/* One scheduler step on the latent: pred = nega + cfg*(posi - nega) (pipelines/wan_video.py:302);
 * out = latents + pred*dsigma (FlowMatchScheduler.step, diffusion/flow_match.py:144-154), each op rounded
 * to bf16 like the reference's tensor arithmetic; nega==NULL means cfg_scale==1 (pred = posi).
 * out may alias latents. */
int fg_cfg_euler_bf16(const void* latents, const void* posi, const void* nega, void* out,
                      int64_t n, float cfg_scale, float dsigma, fg_stream_t stream);

/* ------------------------------------------------------------------------------- VAE decode kernels
 * Channels-last activations: (T,H,W,C) bf16.  */

/* RMS_norm over channels (+ optional SiLU): models/wan_video_vae.py:55-70 and the nn.SiLU after it
 * (:275-279,885): y = silu(bf16(bf16(bf16(x/max(||x||,1e-12)) * sqrt(C)) * gamma)). */
int fg_vae_rmsnorm_silu_bf16(const void* x, const void* gamma, void* out,
                             int64_t pixels, int C, int apply_silu, fg_stream_t stream);

/* Repack a Conv3d/Conv2d weight (Cout,Cin,kt,kh,kw) bf16 -> [kt*kh*kw][Cout_pad][Cin_pad] bf16 with
 * Cin_pad = roundup(Cin,64), Cout_pad = roundup(Cout,128) (zero filled).  fg_conv_packed_bytes gives the size. */
int64_t fg_conv_packed_bytes(int Cout, int Cin, int kt, int kh, int kw);
int fg_conv_pack_weight_bf16(const void* w, void* packed, int Cout, int Cin, int kt, int kh, int kw,
                             fg_stream_t stream);

/* Causal conv as implicit GEMM on MFMA: CausalConv3d.forward models/wan_video_vae.py:33-52 (kt in {1,3},
 * kh==kw in {1,3}, stride 1, "same" spatial zero padding, causal time padding); resample == 1: the Conv2d after
 * nearest-exact 2x upsampling in Resample38 (:242-251; input is (.,H/2,W/2,Cin)); resample == 2: the encoder's
 * ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2) of Resample38 downsample (:253-263; input is (.,2H,2W,Cin)).
 * x: (T + kt-1, Hin, Win, Cin) — the first kt-1 frames are the layer's feature cache, i.e. the previous input
 * frames that the reference concatenates in front (:46-49), zeros on the first chunk / after 'Rep'; output frame
 * t reads input frames t .. t+kt-1.  out (T,H,W,Cout), or with time_interleave!=0 (Resample.forward :153-156,
 * Cout = 2*Cout2): out (2T,H,W,Cout2) with channel block j of frame t written to frame 2t+j.
 * residual (same shape as out, or NULL) is added after the bf16 rounding of conv+bias (ResidualBlock :301).
 * Cin % 8 == 0; Cout % 4 == 0; input and packed weights < 3.75 GiB each (32-bit buffer offsets). */
int fg_conv3d_cl_bf16(const void* x, const void* w_packed, const void* bias,
                      const void* residual, void* out,
                      int T, int H, int W, int Cin, int Cout, int kt, int ks,
                      int resample, int time_interleave, fg_stream_t stream);
/* Diagnostic: the tile variant fg_conv3d_cl_bf16 launches for an output of (T,H,W,Cout): 256 = conv3d_cl_256_kernel
 * (256x256x64 tile, LDS-DMA staging) when it fills the chip, else 128 = conv3d_cl_kernel (128x128x64). */
int fg_conv_tile_choice(int T, int H, int W, int Cout);

/* out = main + DupUp3D(x): models/wan_video_vae.py:417-439,510-512.  x (T,H,W,Cin); main/out
 * (T*ft - drop, H*fs, W*fs, Cout) where drop = ft-1 if first_chunk. */
int fg_dupup3d_add_bf16(const void* x, const void* main_path, void* out,
                        int T, int H, int W, int Cin, int Cout, int ft, int fs, int first_chunk,
                        fg_stream_t stream);

/* Row softmax of fp32 scores * scale -> bf16 probabilities (VAE mid AttentionBlock,
 * models/wan_video_vae.py:331-336; its two GEMMs stay on hipBLASLt). */
int fg_softmax_rows_f32_bf16(const float* scores, void* probs, int64_t rows, int64_t cols, float scale,
                             fg_stream_t stream);

/* Layout boundary: latent NCTHW (C,T,H,W) -> channels-last (T,H,W,C) with the de-normalisation
 * z/scale[1] + scale[0] of VideoVAE38_.decode (models/wan_video_vae.py:1328-1331): mean/inv_std are bf16 (C). */
int fg_vae_latent_to_cl_bf16(const void* z, const void* mean, const void* inv_std, void* out,
                             int C, int T, int H, int W, fg_stream_t stream);

/* Decoder head output (T,H,W,12) channels-last -> unpatchify(…,2) (models/wan_video_vae.py:214-224,1349)
 * -> frames [t0, t0+T) of a (3,F,2H,2W) NCTHW video; clamp to [-1,1] when do_clamp
 * (single_decode :1215). */
int fg_vae_unpatchify_bf16(const void* x, void* video, int T, int H, int W, int F, int t0, int do_clamp,
                           fg_stream_t stream);

/* Tiled-decode feathering (WanVideoVAE.tiled_decode models/wan_video_vae.py:1124-1149, masks :1081-1100):
 * values[:, :, y0:y0+th, x0:x0+tw] += tile*mask ; weight[...] += mask, bf16 accumulators like the
 * reference; mask(y,x) = min(ramp_h(y), ramp_w(x)), ramps of width border_h/border_w on non-bound sides.
 * values (C,F,Hv,Wv), weight (F,Hv,Wv), tile (C,F,th,tw) (C = 3 decode, 48 tiled_encode :1176-1201).
 * bound bits: 1 top, 2 bottom, 4 left, 8 right. */
int fg_vae_tile_accumulate_bf16(const void* tile, void* values, void* weight,
                                int C, int F, int Hv, int Wv, int th, int tw, int y0, int x0,
                                int border_h, int border_w, int bound_bits, fg_stream_t stream);

/* values = values / weight, clamped to [-1,1] when do_clamp (decode :1150-1151; encode :1202 does not clamp). */
int fg_vae_tile_finalize_bf16(void* values, const void* weight, int C, int F, int Hv, int Wv, int do_clamp,
                              fg_stream_t stream);

/* Encoder boundary (first-frame conditioning, VideoVAE38_.encode models/wan_video_vae.py:1298-1323):
 * video (3,T,H,W) NCTHW -> patchify(.,2) (:199-211) -> channels-last (T,H/2,W/2,16): 12 channels + 4 zeros, so the
 * 12->dim conv1 reads whole 16-byte chunks (its packed weights are zero beyond channel 12). */
int fg_vae_patchify_bf16(const void* video, void* out, int T, int H, int W, fg_stream_t stream);

/* out = main + AvgDown3D(x): Down_ResidualBlock shortcut (models/wan_video_vae.py:363-395,474).  x (T,H,W,Cin);
 * main/out (ceil(T/ft),H/fs,W/fs,Cout); a missing leading frame counts as zeros (the reference pads in front). */
int fg_avgdown3d_add_bf16(const void* x, const void* main_path, void* out,
                          int T, int H, int W, int Cin, int Cout, int ft, int fs, fg_stream_t stream);

/* Encoder tail (:1314-1321): x (T,H,W,Cx) channels-last, mu = first Z channels; out (Z,T,H,W) = (mu - mean) * inv_std. */
int fg_vae_latent_from_cl_bf16(const void* x, const void* mean, const void* inv_std, void* out,
                               int Z, int Cx, int T, int H, int W, fg_stream_t stream);

/* (3,F,H,W) bf16 in [-1,1] -> (F,H,W,3) uint8 by ((x+1)*127.5).clip(0,255) truncation
 * (BasePipeline.vae_output_to_video, diffusion/base_pipeline.py:128-143). */
int fg_video_to_uint8(const void* video, void* out_u8, int F, int H, int W, fg_stream_t stream);

/* ------------------------------------------------------------------------ umT5 text encoder helpers
 * (WanTextEncoder runs twice per clip before the loop; its GEMMs stay on hipBLASLt, T5LayerNorm is
 * fg_rmsnorm_rope_bf16 without tables.) */

/* T5Attention softmax, models/wan_video_text_encoder.py:72-87: probs = softmax over keys of
 * bf16(scores + bias), bias replaced by finfo(bf16).min where key_mask[c] == 0 (key_mask may be NULL);
 * scores/bias/probs (rows, cols) bf16, fp32 softmax, no 1/sqrt(d) scaling. */
int fg_softmax_bias_bf16(const void* scores, const void* bias, const int* key_mask, void* probs,
                         int64_t rows, int64_t cols, fg_stream_t stream);

/* T5FeedForward gate, :19-22,109-111: out = fc1 * (0.5*g*(1+tanh(sqrt(2/pi)*(g+0.044715*g^3)))), each tensor op of
 * that expression rounded to bf16 as the reference's explicit GELU module does. */
int fg_gated_gelu_bf16(const void* fc1, const void* gate, void* out, int64_t n, fg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FAIRYGEN_HIP_H */
