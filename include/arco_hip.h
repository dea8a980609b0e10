/* arco_hip.h - C ABI of libarco_hip.so: the MI355X (gfx950) kernels behind the ARCO
 * stratified pixel-contrastive training hot path.
 *
 * The reference (charlesyou999648/ARCO) is pure Python/PyTorch: it has no FFI layer, so the
 * "interface each entry point replaces" is the PyTorch call sequence at the cited reference
 * file:line (paths relative to the reference's code/ directory).  The Python host
 * (arco_amd/*.py) binds these with ctypes (arco_amd/_lib.py); INTEGRATION.md shows the stub.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *    (PyTorch allocates), except the two host-side sampler entry points at the end;
 *  - no allocation, no ownership transfer, no global state: kernels are stateless, re-entrant,
 *    and are enqueued on the caller's HIP stream (`stream` = hipStream_t);
 *  - activations are channels-last fp32 "rows": pixel p of an [N, C, *spatial] tensor is the C
 *    consecutive floats at base + p*ld (ld >= C lets a channel slice be used in place);
 *  - return 0 on success, <0 on error (-1 bad argument, -2 launch failure, -3 unsupported);
 *    the *_ws_* / *_blocks / *_bytes helpers return sizes.
 */
#ifndef ARCO_HIP_H
#define ARCO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- L1/L2  per-pixel class masks, counts, stable compaction (loss_helper_3d.py:341-342,352-358,
 *      364-401; loss_helper.py:512-572).  codes: bit c = low-valid, 21+c = anchor candidate,
 *      42+c = negative key (C <= 21).  totals = [n_low_valid[C] | n_anchor[C] | n_neg[C]].           */
int arco_mask_codes(const int64_t* lab_l, const int64_t* lab_u, const float* prob_l, const float* prob_u,
                    const float* low_mask, const float* high_mask, int n_l_img, int n_u_img, int C, long P,
                    float delta_p, float delta_n, int low_rank, int high_rank, uint64_t* codes,
                    uint32_t* block_counts, uint32_t* block_offsets, int64_t* totals, void* stream);
/* lists[(kind*C + c)*n_pix + j] = row id of the j-th anchor (kind 0) / negative (kind 1) pixel of class c,
 * in (b, spatial) row-major order == torch boolean-mask order (loss_helper_3d.py:377,403).                */
int arco_compact_rows(const uint64_t* codes, long n_pix, int C, const uint32_t* block_offsets, int32_t* lists,
                      void* stream);
/* prototype[c] = mean of teacher rows over low-valid pixels (loss_helper_3d.py:380-384); NaN when empty. */
long arco_proto_ws_floats(long n_pix, int C, int D);
int arco_masked_proto(const float* T, long ldt, const uint64_t* codes, long n_pix, int C, int D,
                      const int64_t* totals, float* partial, float* proto, void* stream);
/* linear-prototype path: low-valid bits as float rows; weighted row sums sum_rows Wt[row][c]*T[row][:]
 * (prototype = W_fea4 . masked mean of the fea4 INPUT, mask pushed through the bilinear adjoint)           */
int arco_lv_weights(const uint64_t* codes, long n_pix, int C, int Cp, float* W, void* stream);
int arco_weighted_row_sum(const float* T, long ldt, const float* Wt, long ldw, long n_rows, int C, int D,
                          const int64_t* totals, float* partial, float* out, long ldo, void* stream);
/* out[j] = src[list ? list[idx[j]] : idx[j]], idx = idx64 | idx32 | identity, j in [first, first+n)
 * (rep[mask][idx] / rep_teacher[negative_mask], loss_helper_3d.py:403,455-457).                           */
int arco_gather_rows(const float* src, long ld_src, int D, const int32_t* list, const int64_t* idx64,
                     const int32_t* idx32, long first, long n, float* out, long ld_out, void* stream);
/* the same two with an f16 row matrix as the source (BASELINE.json configs[4], --act_dtype f16: the V-Net's full-resolution
 * feature maps are consumed by the heads as stored - FeatureExtractor_3d rows model_3D.py:52-58 - sums and rows are fp32)    */
int arco_weighted_row_sum_h(const void* T, long ldt, const float* Wt, long ldw, long n_rows, int C, int D,
                            const int64_t* totals, float* partial, float* out, long ldo, void* stream);
int arco_gather_rows_h(const void* src, long ld_src, int D, const int32_t* list, const int64_t* idx64,
                       const int32_t* idx32, long first, long n, float* out, long ld_out, void* stream);
/* ---- L3  out = cat(old, keys)[-min(len_old+n, queue_size):]  (dequeue_and_enqueue, loss_helper_3d.py:12-32) */
int arco_bank_append(const float* old, long len_old, const float* keys, long n, long queue_size, int D, float* out,
                     void* stream);
/* ---- L6  InfoNCE (loss_helper_3d.py:503-509): y = x/max(||x||,eps) (+transposed copy, +1/norm) ...      */
int arco_normalize_rows(const float* x, long ldx, long n, int D, float eps, float* y, long ldy, float* yt, long ldyt,
                        float* inv, void* stream);
/* ... M[q][k] = multiplicity of bank row k among query q's sampled negatives ...                          */
int arco_neg_multiplicity(const int64_t* idx, int Q, int Nn, long L, long ld, uint32_t* M, void* stream);
/* ... loss_q = logsumexp([pos, S[q, idx]]/T) - pos/T ; W = d loss_q/dS ; gpos = d loss_q/dpos             */
int arco_infonce_fwd(const float* S, long ld, const uint32_t* M, long L, const float* An, const float* Pn, long ldp,
                     int Q, int D, float temp, float* W, float* gpos, float* loss_q, void* stream);
/* ... dA = scale * d(loss)/dA through the cosine normalisation (G = W @ Bn)                                */
int arco_infonce_anchor_grad(const float* G, const float* An, const float* Pn, long ldp, const float* gpos,
                             const float* inv, int Q, int D, float eps, float scale, float* dA, void* stream);
/* ---- L5-L6 grouped: every class ("entry") of the per-class loop of loss_helper_3d.py:435-509 in ONE launch per stage.
   Host arrays of E <= 21 entries: banks[e] (device pointer of memobank[valid_classes[k]][0]), lens[e] (its rows),
   prow[e] (row of seg_proto[k] in the normalised prototype matrix), k[e] (loop counter = row-list index).              */
/* y = x / max(||x||, eps) written with zeroed pad columns [D, Dp) (anchors, prototypes)                                 */
int arco_normalize_rows_pad(const float* x, long ldx, long n, int D, int Dp, float eps, float* y, long ldy, float* inv,
                            void* stream);
/* Bn[E][Lp][Dp] = normalised bank rows (zero beyond lens[e] / D), Bt[E][Dp][Lp] = their transposes (nullable)          */
int arco_nce_normalize_banks(const void* const* banks, const int* lens, int E, int D, int Dp, long Lp, float eps, float* Bn,
                             float* Bt, void* stream);
/* batch independent GEMMs out_z = in_z . W_z^T (operand z at base + z*stride floats), optional split-K through ws:
   the cosine scores A_c . Bank_c^T (loss_helper_3d.py:503-505) and the anchor-gradient GEMMs W_c . Bank_c              */
int arco_gemm_batched(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, long M,
                      int batch, long stride_in, long stride_w, long stride_out, int splits, float* ws, void* stream);
/* per (query, entry): multiplicities of the sampled negatives (LDS), loss_q = logsumexp([pos, S[q, idx]]/T) - pos/T,
   W = d loss_q / dS over the whole padded row, gpos = d loss_q / d pos (loss_helper_3d.py:478-509)                      */
long arco_nce_max_len(void);
int arco_nce_fused(const float* S, long ld, const int* lens, const int* prow, int E, const int64_t* idx_all, long idx_off,
                   long idx_stride, int Q, int Nn, const float* An, const float* Pn_all, int Dp, float temp, float* W,
                   float* gpos, float* loss_q, void* stream);
/* dA[e*Q+q][0..D) = scale * d loss / d anchor through the cosine normalisation, all entries                            */
int arco_nce_anchor_grad(const float* G, const float* An, const float* Pn_all, const int* prow, int E, const float* gpos,
                         const float* inv, int Q, int D, int Dp, float eps, float scale, float* dA, long ld_dA, void* stream);
/* ---- L6, round 6: the score GEMM with the temperature-scaled softmax-CE in its EPILOGUE (north_star's wording; loss_helper_3d.py:503-509:
 *   logits = cat(pos, cosine(anchor, negatives)) / T; F.cross_entropy(logits, 0)) - no score matrix in HBM, the bank normalisation
 *   folded into the GEMM's B staging.  Five launches replace nine (2 x normalize_rows_pad, nce_normalize_banks, gemm_batched, nce_fused,
 *   sum_scale | gemm_batched + slab sum, nce_anchor_grad):
 *   arco_nce_prep    anchors / prototypes normalised (An, invA, Pn) + per (entry, query) the uint16 multiplicity row M[e*Q+q][0..Lp)
 *   arco_nce_score   S = An . bank^T on the matrix cores per (64 queries, 128 bank rows, entry); the B staging accumulates each bank
 *                    row's sum of squares; epilogue Wu = M * exp((cos - 1)/T) / ||b||, partial row sums Zp; pos = An . Pn[prow];
 *                    Bt = the raw bank transposed (the operand of the anchor-gradient GEMM)
 *   arco_nce_finish  loss_q = log Z - (pos - 1)/T with Z = sum Zp + exp((pos - 1)/T); gpos, gscale = 1/(T Z); the loss sum
 *   then arco_gemm_batched(Wu, Bt) and arco_nce_anchor_grad_scaled.  Same results as the staged route to fp32 rounding.         */
int arco_nce_prep(const float* A, long n_a, const float* P, long n_p, int D, int Dp, float eps, float* An, float* invA, float* Pn,
                  const int* lens, int E, const int64_t* idx_all, long idx_off, long idx_stride, int Q, int Nn, long Lp, void* M,
                  void* stream);
long arco_nce_score_ltiles(long Lp);
int arco_nce_score(const float* An, int Dp, int D, const void* const* banks, const int* lens, const int* prow, int E, long Lp, int Q,
                   const void* M, const float* Pn_all, float temp, float eps, float* Wu, float* Zp, float* pos, float* Bt, void* stream);
int arco_nce_finish(const float* pos, long n_rows, const float* Zp, long Lp, float temp, float scale, float* gpos, float* gscale,
                    float* loss_q, float* loss_sum, void* stream);
int arco_nce_anchor_grad_scaled(const float* Gu, const float* An, const float* Pn_all, const int* prow, int E, const float* gpos,
                                const float* inv, const float* gscale, int Q, int D, int Dp, float eps, float scale, float* dA, long ld_dA,
                                void* stream);
/* out[e*Q+q] = lists[k[e]][idx_all[e*idx_stride + q]]: pixel id of every sampled anchor (loss_helper_3d.py:455-457)     */
int arco_anchor_pix(const int32_t* lists, long n_pix, const int* k, int E, const int64_t* idx_all, long idx_stride, int Q,
                    int64_t* out, void* stream);
/* dst[list?list[idx[j]]:idx[j]] += alpha * (alpha_dev?*alpha_dev:1) * src[j]  (backward of the anchor gather) */
int arco_scatter_add_rows(const float* src, long ld_src, int D, const int32_t* list, const int64_t* idx, long n,
                          const float* alpha_dev, float alpha, float* dst, long ld_dst, void* stream);
int arco_sum_scale(const float* x, int n, float scale, float* out, int accumulate, void* stream);

/* 1 when a conv of this shape runs on the split-bf16 kernels (mma = 3: every fp32 operand = 3 bf16 terms, six bf16 MFMAs
   per 32 k, fp32-accurate) and therefore takes the split-packed weights (arco_pack_conv_weight mode | 2); else 0      */
int arco_conv_split_ok(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in);
/* ---- N1-N4  convolutions on the fp32 matrix cores (nn.Conv2d 3x3 / 1x1: unetWithArgs.py:36-44,72,139;
 *      model_2D.py:25-33; train_arco_2d.py:231-234).  Wp = packed weights [taps][ceil16(N)][ceil16(K)].    */
int arco_pack_conv_weight(const float* W, int Cout, int Cin, int taps, int mode, float* Wp, void* stream);
/* all conv weights of a model in ONE launch: `desc` = device array of records {const float* src; float* dst;
 * int Cout, Cin, taps, mode, Npad, Kpad; long first;} (arco_pack_desc_bytes() bytes each).
 * mode bits: 1 data-gradient form (flipped + transposed), 2 split-bf16, 4 f16; mode >> 3 != 0 gathers the GEMM form of a
 * k2 s2 (transposed) convolution straight from the torch layout - nn.Conv3d(k2, s2) / nn.ConvTranspose3d(k2, s2) of
 * vnetWithArgs.py:67-118: 1 = W2[co][t*ci + c] = W[co][c][t], 2 = W2[t*co + o][ci] = W[ci][o][t], 4 = the bias repeated over
 * the 8 taps; `taps` then carries the inner dimension (ci resp. co), Cout / Cin are W2's logical [N][K]                */
long arco_pack_desc_bytes();
int arco_pack_many(const void* desc, int n_desc, long total, void* stream);
/* small-M x N, long-K GEMM (InfoNCE anchor gradient, loss_helper_3d.py:503-509 backward): K split into `splits`
 * slabs over grid.y, slab outputs in ws (splits*M*ld_out floats), fixed-order sum into out                  */
int arco_gemm_splitk(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, long M,
                     int splits, float* ws, void* stream);
int arco_conv_mblocks(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups);
int arco_conv_mblocks_mma(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups, int mma);   /* ... for a launch in matrix-core mode mma (arco_conv3d_fwd) */
int arco_conv_mblocks_pro(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int stat_groups, int mma, int pro_groups);   /* ... for a launch through arco_conv3d_fwd_pro (3x3x3: the forms with the activation in their loaders tile differently) */
/* which kernel instantiation a launch uses: igemm_kernel<TAPS,BM,BN,..> -> TAPS*1e6 + BM*1e3 + BN;
 * conv3x3_halo_kernel<CIN,COUT,..> -> 9.9e6 + CIN*1e3 + COUT */
int arco_conv_config(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int* kc_depth_db);
int arco_conv_config_mma(int taps, int NB, int H, int W, int Cin, int Cout, long ld_in, int mma);
/* A/B switch of the software-pipelined split-bf16 3x3 kernel (conv_sp.hip; ids 9.3e6 + A_T*1e3 + BN, resident-weights form 9.35e6 + ...): on = 1 (default,
 * or ARCO_CONV_SP=0 in the environment for off) lets the eligible wide 2-D shapes take it, 0 keeps every shape on
 * igemm_kernel.  Returns the previous setting.  Tile counts differ: query arco_conv_mblocks_mma after switching.   */
int arco_conv_sp_set(int on);
/* A/B switch of the software-pipelined flat-tile 3x3x3 kernel (conv3d_fl.hip; ids 9.27e6 / 9.28e6 / 9.29e6 + A_T*1e3 + BN for its per-step, depth-walking and per-chunk forms: the 32 .. 256-channel levels of
 * the V-Net, vnetWithArgs.py:5-31): on = 0 (or ARCO_CONV3D_FL=0) keeps every 3x3x3 launch on igemm_kernel.  Outputs are bit-identical
 * either way; tile counts differ (query arco_conv_mblocks_mma after switching).  Returns the previous setting. */
int arco_conv3d_fl_set(int on);
/* A/B switch of the software-pipelined split-bf16 1x1 GEMM (gemm_sp.hip; the wide many-tile FeatureExtractor / q_representation GEMMs,
 * model_2D.py:20-55, train_arco_2d.py:231-234): on = 0 (or ARCO_GEMM_SP=0) keeps every GEMM on igemm_kernel; min_tiles > 0 also sets the
 * smallest 128 x 128 tile count the kernel takes (default 2048, ARCO_GEMM_SP_TILES).  Outputs are bit-identical either way.  Returns the
 * previous `on`. */
int arco_gemm_sp_set(int on, long min_tiles);
/* out = W . in + trilinear_align_corners(lo): the 1x1x1 conv over a high-resolution feature map with the upsampled low-resolution
 * product as residual, sampled in the GEMM's epilogue (FeatureExtractor_3d, model_3D.py:46-58: `fea_i(cat(up(x), f_i)) + cat(...)` with
 * the wide weight block evaluated below the upsample).  Wp: split-bf16 pack.  Returns ARCO_ERR_UNSUPPORTED (-3) when the pipelined
 * kernel does not take the shape - the caller then upsamples with arco_trilinear_fwd and passes the result as arco_conv3d_fwd's residual;
 * both routes give bit-identical outputs. */
int arco_conv1x1_upres_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out, const float* lo,
                           long ld_lo, int NV, int uD, int uH, int uW, int oD, int oH, int oW, void* stream);
/* out = conv(in) (+bias)(+residual); optional per-channel (sum, sumsq) block partials for train-mode BN. */
int arco_conv_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                  const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                  int NB, int H, int W, void* stream);
/* 3-D: NV volumes of D3 planes of H x W; taps = 27 is nn.Conv3d(3, padding=1) (vnetWithArgs.py:16)      */
int arco_conv3d_fwd(const float* in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                    const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                    int NV, int D3, int H, int W,
                    int stat_groups /* BN groups: volumes [g*NV/G,..) feed stat slabs [g*nmb/G,..); 1 = one batch */, int mma /* 0: fp32 MFMA (default, parity); 1 / 2: 3x3x3 operands rounded to f16 / bf16 in registers, fp32 accumulate
                               (BASELINE.json configs[4] "fp16 MFMA conv"; tolerance 1e-2; opt-in) */,
                    void* stream);
int arco_conv3d_wgrad(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NV,
                      int D3, int H, int W, float* ws, float* dW, int accumulate,
                      int mma /* 0 exact fp32; != 0: 3x3x3 halo kernels with bf16 MFMA operands, fp32 accumulate */, void* stream);
long arco_wgrad_ws_floats(int Cout, int Cin, int taps, long M);

/* ---- Consumer-side activation: a block's first Conv-BN-LeakyReLU-Dropout stage (unetWithArgs.py:36-44: conv_conv[0..3]) without the
 * BatchNorm-apply pass.  The producing convolution writes its PRE-activation z (and the BN partial statistics); the block's second
 * convolution - and, in the backward pass, the weight gradient of that second convolution - form
 *     a = dropout(lrelu((z - mean) * istd * gamma + beta))
 * element by element in their loaders while they stage z for the matrix cores (conv_sp.hip producer waves, wgrad_split_kernel): the
 * activation never exists in HBM (2 of the 4 HBM crossings of every such activation, and one launch per stage, are gone).  The
 * arithmetic is arco_bn_act_fwd's operation for operation: results are bit-identical to the two-pass route.                        */
typedef struct ArcoActPro {
  const float* mean; const float* istd;   /* [groups][K] batch statistics of the producing layer (arco_bn_finalize)                 */
  const float* gamma; const float* beta;  /* [K] nn.BatchNorm2d weight / bias                                                       */
  float slope;                            /* nn.LeakyReLU negative_slope (0: ReLU)                                                   */
  int groups;                             /* BatchNorm groups: images [g*NV/G, (g+1)*NV/G) use statistics row g                     */
  int drop_mode; float p;                 /* 0: none; 1: nn.Dropout(p), the stateless mask of arco_bn_act_fwd (element = pixel*K + k) */
  unsigned long long seed; const unsigned long long* seed_dev;   /* as arco_bn_act_fwd (seed_dev: per-replay salt of a HIP graph)    */
} ArcoActPro;
/* 1 when the entry points below take the shape (taps = 9: 3x3, split-bf16 mode 3, the pipelined kernels - forward and weight gradient;
 * taps = 27: the V-Net's 3x3x3 stage -> stage links (vnetWithArgs.py:5-31; conv3d_fc_kernel's loaders, BatchNorm + ReLU, no dropout) -
 * FORWARD ONLY, for gradient-free passes, with the BatchNorm slab count from arco_conv_mblocks_pro); else the caller runs
 * arco_bn_act_fwd and the plain entry points */
int arco_conv_pro_ok(int taps, int NV, int D3, int H, int W, int Cin, int Cout, long ld_in, int mma, int groups);
int arco_conv3d_fwd_pro(const float* z_in, long ld_in, int K, const float* Wp, int N, float* out, long ld_out,
                        const float* bias, const float* residual, long ld_res, float* stat_sum, float* stat_sq, int taps,
                        int NV, int D3, int H, int W, int stat_groups, int mma, const ArcoActPro* pro, void* stream);
int arco_conv3d_wgrad_pro(const float* dZ, long ld_dz, int Cout, const float* z_in, long ld_in, int Cin, int taps, int NV,
                          int D3, int H, int W, float* ws, float* dW, int accumulate, int mma, const ArcoActPro* pro, void* stream);
/* dW[co][ci][tap] (+)= sum_pix dZ[pix][co] * in[pix+tap][ci]   (torch weight layout)                      */
int arco_conv_wgrad(const float* dZ, long ld_dz, int Cout, const float* in, long ld_in, int Cin, int taps, int NB,
                    int H, int W, float* ws, float* dW, int accumulate, void* stream);
int arco_colsum(const float* X, long ldx, long M, int C, float* ws, float* out, int accumulate, void* stream);
int arco_transpose2d(const float* x, long ldx, int rows, int cols, float* y, long ldy, void* stream);

/* ---- N1  train-mode BatchNorm + (Leaky)ReLU + dropout (nn.BatchNorm2d/LeakyReLU/Dropout,
 *      unetWithArgs.py:36-44; vnetWithArgs.py:16-25)                                                      */
int arco_bn_finalize(const float* ssum, const float* ssq, int nblk, int C, long count, float eps, float momentum,
                     float* mean, float* istd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     int groups /* mean/istd rows: [groups][C] */,
                     int defer_from, float* deferred /* nullable: groups >= defer_from postpone their running-statistics
                     update into deferred[groups - defer_from][2][C] (+1 flag float), see arco_bn_apply_deferred */,
                     void* stream);
/*      postponed momentum updates of many BN layers, one launch: desc = device array of n_layers records
 *      {float* running_mean, running_var, deferred; int C, n; float momentum; int pad} (arco_bn_defer_desc_bytes() each).
 *      The reference runs model(l), model(cj2_l), model(u) (train_arco_2d.py:310-312); the build runs (l, u) as one
 *      grouped pass and cj2_l afterwards - the running statistics must still receive the updates in the reference's order */
long arco_bn_defer_desc_bytes(void);
int arco_bn_apply_deferred(const void* desc, int n_layers, void* stream);
int arco_chan_stats_blocks(long M);
int arco_chan_stats(const float* X, long ldx, long M, int C, float* ssum, float* ssq, int groups, void* stream);
int arco_bn_act_fwd(const float* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                    const float* beta, float slope, int drop_mode, float p, uint64_t seed, long P, float* A, long lda,
                    const uint64_t* seed_dev, int groups, void* stream);
int arco_bn_act_bwd(const float* dA, long ldd, const float* Z, long ldz, long M, int C, const float* mean,
                    const float* istd, const float* gamma, const float* beta, float slope, int drop_mode, float p,
                    uint64_t seed, long P, float* ws, float* dgamma, float* dbeta, int accumulate, float* dZ, long ldo,
                    const uint64_t* seed_dev /* same device salt as the forward (graph replays), or NULL */, int groups, void* stream);
/* ---- N2/N3  nn.MaxPool2d(2) (unetWithArgs.py:55-58); nn.Upsample(bilinear, align_corners=True)
 *      (unetWithArgs.py:74-75, model_2D.py:43-52)                                                          */
/* GroupNorm / InstanceNorm + ReLU of the V-Net blocks (vnetWithArgs.py:19-22,48-51,76-79 `normalization='groupnorm'` =
 * nn.GroupNorm(16, C), `'instancenorm'` = nn.InstanceNorm3d(C)): statistics per sample over a set of cpg channels.
 * Forward: arco_chan_stats(groups = N) -> arco_gn_finalize -> arco_bn_act_fwd(groups = N) (mean / istd rows [N][C]);
 * backward: arco_gn_act_bwd (ws as for arco_bn_act_bwd with groups = N)                                                */
int arco_gn_finalize(const float* ssum, const float* ssq, int nblk, int C, int cpg, int N, long count_per_sample_channel,
                     float eps, float* mean, float* istd, void* stream);
int arco_gn_act_bwd(const float* dA, long ldd, const float* Z, long ldz, long M, int C, const float* mean,
                    const float* istd, const float* gamma, const float* beta, float slope, float* ws, float* dgamma,
                    float* dbeta, int accumulate, float* dZ, long ldo, int N, int cpg, void* stream);
int arco_maxpool2_fwd(const float* X, long ldx, int NB, int H, int W, int C, float* Y, long ldy, void* stream);
/* A = lrelu(BN(Z)) (mean / istd rows [groups][C] as arco_bn_act_fwd, no dropout) and P = maxpool2(A) in one pass: the last
   stage of a ConvBlock whose output feeds the next DownBlock's nn.MaxPool2d and the decoder's skip (unetWithArgs.py:36-44,
   55-58,109-116)                                                                                              */
int arco_bn_act_pool_fwd(const float* Z, long ldz, int NB, int H, int W, int C, const float* mean, const float* istd,
                         const float* gamma, const float* beta, float slope, float* A, long lda, float* P, long ldp,
                         int groups, void* stream);
int arco_maxpool2_bwd(const float* X, long ldx, int NB, int H, int W, int C, const float* dY, long ldy, float* dX,
                      long ldo, void* stream);
/* ... fused with the gradient x receives from its second consumer (unetWithArgs.py:109-116,142-158: x_i feeds both the
   next DownBlock and the decoder's skip concat): dX = maxpool2_bwd(dY) + add                                          */
int arco_maxpool2_bwd_add(const float* X, long ldx, int NB, int H, int W, int C, const float* dY, long ldy, const float* add,
                          long ld_add, float* dX, long ldo, void* stream);
int arco_bilinear_fwd(const float* X, long ldx, int NB, int Hi, int Wi, int C, int Ho, int Wo, float* Y, long ldy,
                      void* stream);
int arco_bilinear_bwd(const float* dY, long ldy, int NB, int Hi, int Wi, int C, int Ho, int Wo, float* dX, long ldx,
                      int accumulate, void* stream);
/* rows of cat(upsample(lo), hi) at selected high-res pixels, and the adjoint scatter (row-sparse head of
 * FeatureExtractor.fea4 / q_representation: model_2D.py:51-53, train_arco_2d.py:324-325)                   */
int arco_gather_upcat_rows(const float* lo, long ldlo, int Clo, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                           int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream);
int arco_scatter_upcat_rows(const float* dX, long ldx, const int64_t* pix, long n, float* dlo, long ldlo, int Clo,
                            int Hi, int Wi, float* dhi, long ldhi, int Chi, int Ho, int Wo, void* stream);
/* second level of the row-sparse head: explicit low-res neighbour rows of each anchor (ids + (ly,lx)),
 * their 4-way lerp + cat with the high-res map, and the adjoint                                            */
int arco_up_neighbors(const int64_t* pix, long n, int Hi, int Wi, int Ho, int Wo, int64_t* nb4, float* lylx, void* stream);
int arco_lerp4_cat_rows(const float* V, long ldv, int Clo, const float* lylx, const float* hi, long ldhi, int Chi,
                        const int64_t* pix, long n, float* X, long ldx, void* stream);
int arco_lerp4_cat_rows_bwd(const float* dX, long ldx, int Clo, const float* lylx, const int64_t* pix, long n, float* dV,
                            long ldv, float* dhi, long ldhi, int Chi, void* stream);
/* V-Net k2s2 (transposed) convs as GEMMs over packed 2x2x2 blocks (vnetWithArgs.py:67-118); trilinear
 * align_corners resize of FeatureExtractor_3d (model_3D.py:46-58)                                          */
int arco_s2d3(float* V, long ldv, int NV, int X2, int Y2, int Z2, int C, float* P, long ldp, int dir, void* stream);
int arco_trilinear_fwd(const float* X, long ldx, int NV, int Di, int Hi, int Wi, int C, int Do, int Ho, int Wo, float* Y,
                       long ldy, void* stream);
int arco_trilinear_bwd(const float* dY, long ldy, int NV, int Di, int Hi, int Wi, int C, int Do, int Ho, int Wo, float* dX,
                       long ldx, void* stream);
/* 3-D row-sparse head: rows of cat(trilinear_up(lo), hi) at selected voxels + adjoint (model_3D.py:52-55) */
int arco_gather_upcat_rows3d(const float* lo, long ldlo, int Clo, int Di, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                             int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream);
int arco_scatter_upcat_rows3d(const float* dX, long ldx, const int64_t* pix, long n, float* dlo, long ldlo, int Clo, int Di,
                              int Hi, int Wi, float* dhi, long ldhi, int Chi, int Do, int Ho, int Wo, void* stream);
/* ... with `hi` stored as f16; and the way back for its row-sparse gradient: rows idx[] of an fp32 buffer -> the same rows of an
 * f16 buffer, times the loss scale, saturated at +-65504 (arco_cast_f2h on the touched rows only); arco_zero_rows for f16       */
int arco_gather_upcat_rows3d_h(const float* lo, long ldlo, int Clo, int Di, int Hi, int Wi, const void* hi, long ldhi, int Chi,
                               int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream);
/* The level below evaluated lazily too (arco_amd/head.py LazyHead3dL3Fn; model_3D.py:46-58 one level further down): V = the eight
 * corner rows (8 j + k, arco_corner_rows3d's order) of every sampled voxel, each already through its own layer; X[j] = cat(their
 * trilinear blend in the gather's arithmetic, hi[pix[j]]); _bwd: dV[8 j + k] = w8[8 j + k] * dX[j][0..Clo)                        */
int arco_lerp8_cat_rows3d(const float* V, long ldv, int Clo, int Di, int Hi, int Wi, const float* hi, long ldhi, int Chi,
                          int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream);
int arco_lerp8_cat_rows3d_h(const float* V, long ldv, int Clo, int Di, int Hi, int Wi, const void* hi, long ldhi, int Chi,
                            int Do, int Ho, int Wo, const int64_t* pix, long n, float* X, long ldx, void* stream);
int arco_lerp8_rows3d_bwd(const float* dX, long ldx, int Clo, const float* w8, long n, float* dV, long ldv, void* stream);
int arco_cast_rows_f2h(const float* src, long ld_src, int C, const int64_t* idx, long n, float scale, void* dst, long ld_dst,
                       void* stream);
int arco_zero_rows_h(void* dst, long ld, int C, const int64_t* idx, long n, void* stream);
/* Order-independent form of the row-scatter adjoints above (csrc/det_scatter.hip): fp32 atomics sum in arrival order (one ulp of
 * run-to-run difference, as in torch's own upsample backward on the GPU that the reference runs: model_3D.py:52-55); these
 * accumulate fixed-point int64 (2^44 units for the largest |source element|, arco_det_absmax) - bit-reproducible.
 *   arco_det_absmax:        maxbits[0] = max bits(|X[r][c]|) over [n, C]  (>= 0x7f800000: a non-finite element)
 *   arco_det_scatter_rows:  acc[r(e)][c] += fix(w[e] * src[e / div][c]),  r(e) = list ? list[idx[e]] : idx[e],  e < n_e  (w nullable = 1)
 *   arco_det_finish_rows:   dst[r(e)][c]  = alpha * fp32(acc[r(e)][c])    (nan when the source was non-finite)
 *   arco_det_clear_rows:    acc[r(e)][c]  = 0                             (acc: persistent, zero between uses)
 *   arco_corner_rows3d:     idx8[8j+k], w8[8j+k] = low-resolution row and weight of corner k of sampled voxel pix[j]            */
int arco_det_absmax(const float* X, long ld, int C, long n, unsigned* maxbits, void* stream);
int arco_det_scatter_rows(const float* src, long ld_src, int C, int div, const int32_t* list, const int64_t* idx, const float* w,
                          long n_e, long long* acc, long ld_acc, const unsigned* maxbits, void* stream);
int arco_det_finish_rows(const int32_t* list, const int64_t* idx, long n_e, const long long* acc, long ld_acc, int C,
                         const unsigned* maxbits, float alpha, float* dst, long ld_dst, void* stream);
int arco_det_clear_rows(const int32_t* list, const int64_t* idx, long n_e, long long* acc, long ld_acc, int C, void* stream);
int arco_corner_rows3d(const int64_t* pix, long n, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int64_t* idx8, float* w8,
                       void* stream);
/*   arco_corner_rows2d:     the bilinear counterpart (four corners; adjoint of arco_gather_upcat_rows, model_2D.py:43-50)              */
int arco_corner_rows2d(const int64_t* pix, long n, int Hi, int Wi, int Ho, int Wo, int64_t* idx4, float* w4, void* stream);
/* Row-sparse backward of the dense per-pixel layers (nn.Conv2d(k=1) of FeatureExtractor / q_representation called on whole maps:
 * model_2D.py:51-53, train_arco_2d.py:231-234, 324-326).  The loss reads `rep` at a few hundred sampled rows (loss_helper_3d.py:455-457),
 * so d loss / d rep - and, the layers being per-pixel, every gradient down to the first resampling - is zero in all other rows:
 *   arco_row_nonzero: flag[r] = any element of row r that is not +-0   (one pass over the gradient)
 *   arco_put_rows:    dst[idx[j]] = src[j]  (idx unique: the compacted data gradient back into a zero tensor)                          */
int arco_row_nonzero(const float* X, long ld, int C, long M, unsigned char* flag, void* stream);
int arco_put_rows(const float* src, long ld_src, int C, const int64_t* idx, long n, float* dst, long ld_dst, void* stream);
/* glue kernels replacing chains of tensor-library launches in the step (no reference counterpart: the reference's
 * autograd does these as separate zeros / add / copy / mul kernels, train_arco_2d.py:426-431, model_2D.py:43-50):
 * arco_zero_rows: rows idx[] of a [rows, ld] buffer zeroed over C channels (re-arms a persistent gradient buffer);
 * arco_fold_residual / arco_unfold_residual: W' = W + I split into its column blocks [n, c] | [n, n - c], and the gradient back;
 * arco_combine_terms(_bwd): out = sum_i w_i * term_i over <= 8 device scalars (host arrays of pointers / weights), grads_i = w_i * g */
int arco_zero_rows(float* dst, long ld, int C, const int64_t* idx, long n, void* stream);
int arco_fold_residual(const float* W, int n, int c, float* lo, float* hi, void* stream);
int arco_unfold_residual(const float* dlo, const float* dhi, int n, int c, float* dW, void* stream);
int arco_combine_terms(const float* const* terms, const float* weights, int n, float* out, void* stream);
int arco_combine_terms_bwd(const float* weights, int n, const float* g, float* grads, void* stream);
int arco_copy_rows(const float* X, long ldx, long M, int C, float* Y, long ldy, int accumulate, void* stream);
int arco_nchw_to_nhwc(const float* X, int NB, int C, long P, float* Y, long ldy, void* stream);
int arco_nhwc_to_nchw(const float* X, long ldx, int NB, int C, long P, float* Y, void* stream);

/* ---- O1/N5  torch.optim.SGD(nesterov) step and EMA over flat buffers (train_arco_2d.py:248,306-308,431-432;
 *      model_2D.py:176-182)                                                                                */
int arco_sgd_nesterov(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                      int first, void* stream);
/* the same with nesterov=False (stage-1 pre-training: pretrain_2D.py:193-195, pretrain_3D.py) */
int arco_sgd_momentum(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                      int first, void* stream);
int arco_ema(float* k, const float* q, long n, float m, void* stream);

/* ---- T1  trainer glue (train_arco_2d.py:284-286,342-393,492-498)                                        */
int arco_softmax_rows(const float* X, long ld, long M, int C, long P, float* prob_planes, float* maxp, int64_t* amax,
                      float* entropy, void* stream);
int arco_label_onehot(const int64_t* lab, long M, int C, long P, int64_t* out, void* stream);
/* ---- §8f row 1: supervised CrossEntropy + Dice (train_arco_2d.py:336-339, utils/losses.py:173-209) and the
 *      confidence-weighted unsupervised CE (train_arco_2d.py:482-489) on channels-last logits             */
/* ---- E  equivariance loss (train_arco_2d.py:404-423; tps/rand_tps.py, tps_stn_pytorch/tps_grid_gen.py:59-71,
 *      tps/grid_sample.py:11-12 = F.grid_sample(bilinear, align_corners=True))                                  */
int arco_tps_grid(const float* rep, const float* mapping, int B, long HW, int NR, float* grid, void* stream);
int arco_grid_sample_fwd(const float* X, long ldx, int NB, int H, int W, int D3 /* slices per volume, 1 in 2-D */, int C,
                         const float* grid, int Ho, int Wo, int border, float* Y, long ldy, void* stream);
/* ---- A3 photometric augmentation of batch_transform (augment.py:148-225, 255-281): Pillow's 8-bit integer arithmetic
   (to_pil_image quantisation, ImageEnhance blends of torchvision's ColorJitter in fn_idx order, convert L / HSV,
   ImagingGaussianBlur = 3 box passes per direction, to_tensor).  data / out [B,C,H,W] fp32, C = 1 or 3, B <= 32;
   desc: HOST array of B records {int order[4]; float f[4]; int jitter, blur; float blur_r;} (arco_jitter_desc_bytes() each);
   ws: B uint64 + B*C*H*W floats.  arco_quantize8: floor(x*255)/255 (the confidence map's PIL round trip).               */
long arco_jitter_desc_bytes(void);
int arco_jitter_blur(const float* data, int B, int C, int H, int W, const void* desc_host, void* ws, float* out, void* stream);
int arco_quantize8(const float* x, long n, float* out, void* stream);
/* ---- A2 AdvMorph (adv_morph.py:310-580): 2-channel fields kept channels-last [B,H,W,2] (a field is a grid_sample grid;
   applyComposition2D = arco_grid_sample_fwd with border padding).  out = alpha*in + beta*base_grid (get_base_grid, :184-207),
   optional clamp to [-1,1]; depthwise ks x ks filter with zero padding (gaussian_smooth, :445-497; weights: host array);
   F.interpolate(bilinear, align_corners=False) (:507-508)                                                              */
int arco_field_axpb(const float* in, float alpha, float beta, const float* in2, float gamma, int B, int H, int W, int clamp,
                    float* out, void* stream);   /* out = alpha*in + beta*base_grid + gamma*in2 */
int arco_field_smooth(const float* in, int B, int H, int W, int C, int ks, const float* weights_host, float* out, void* stream);
int arco_field_resize(const float* in, int B, int h, int w, int C, int H, int W, float* out, void* stream);
int arco_eqv_loss_fwd(const float* P_, long ldp, const float* Q_, long ldq, const float* mask, int B, long P, int C, double* ws,
                      float* out, void* stream);
int arco_eqv_loss_bwd(const float* P_, long ldp, const float* Q_, long ldq, const float* mask, int B, long P, int C,
                      const double* ws, const float* g, float* dP, long ldo, void* stream);
/* ---- V  evaluation (test_2D.py:52-66): out[c] = {|pred==c|, |gt==c|, |pred==c & gt==c|} as int64[C][3]          */
int arco_overlap_counts(const int64_t* pred, const int64_t* gt, long n, int C, int64_t* out, void* stream);
/* ---- A  mixing strategies of the unlabeled stream (augment.py:284-313 generate_unsup_data, masks :230-252; volumes
 *      augment_3d.py:182-257).  desc_host[B][8] = {y0, y1, x0, x1, z0, z1, sel_lo, sel_hi} in HOST memory (the host
 *      draws it; it travels in the kernel arguments); Z = 1 in 2-D; mode 0 cutmix, 1 cutout, 2 classmix; data NC[spatial] */
int arco_mix_unsup(const float* data, int Cimg, const int64_t* target, const float* logits, int B, int H, int W, int Z,
                   const int* desc_host, int mode, float* odata, int64_t* otarget, float* ologits, void* stream);
/*      presence[i] = bit set of the labels occurring in target[i] (torch.unique of augment.py:248), labels < 64          */
int arco_label_presence(const int64_t* target, int B, long HW, uint64_t* presence, void* stream);
/* ---- V  3-D sliding-window evaluation (test_util.py:139-211): score[C][ww][hh][dd] += prob[C][px][py][pz] at (xs, ys, zs),
 *      cnt += 1 (test_util.py:196-199); then score /= cnt, label = argmax over classes (:200-201)                    */
int arco_window_accumulate(const float* prob, int C, int px, int py, int pz, float* score, float* cnt, int ww, int hh, int dd,
                           int xs, int ys, int zs, void* stream);
int arco_score_finalize(float* score, const float* cnt, int C, long vol, int64_t* label, void* stream);
long arco_seg_ws_doubles(long M, int C, int B);
int arco_sup_loss_fwd(const float* X, long ld, long M, int C, const int64_t* lab, double* ws, float* out, void* stream);
int arco_sup_loss_bwd(const float* X, long ld, long M, int C, const int64_t* lab, const double* ws, const float* g_ce,
                      const float* g_dice, float* dX, long ldo, void* stream);
/* utils/losses.py:173-209 DiceLoss.forward on probability rows [M, C] (what train_arco_2d.py:269,338 /
 * train_arco_3d.py:245,308 instantiate and call): per-class weights or NULL; ws as for arco_sup_loss_fwd                  */
int arco_dice_probs_fwd(const float* P, long ld, long M, int C, const int64_t* lab, const float* wgt, double* ws, float* out,
                        void* stream);
int arco_dice_probs_bwd(const float* P, long ld, long M, int C, const int64_t* lab, const float* wgt, const double* ws,
                        const float* g, float* dP, long ldo, void* stream);
/* workspaces of the two per-image losses: unsup ws >= arco_loss_slabs(B) * 4 * B + B + 1 doubles, eqv ws >= arco_loss_slabs(B) * 2 * B + B */
long arco_loss_slabs(int B);
int arco_unsup_loss_fwd(const float* X, long ld, int B, long P, int C, const int64_t* lab, const float* conf, float thr,
                        double* ws, float* out, void* stream);
int arco_unsup_loss_bwd(const float* X, long ld, int B, long P, int C, const int64_t* lab, const double* ws, const float* g,
                        float* dX, long ldo, void* stream);
long arco_sel_state_bytes();
/* exact np.percentile(entropy[valid], q) (linear) by device radix select -> low/high masks              */
int arco_entropy_masks(const float* ent, const int64_t* lab_l, const int64_t* lab_u, long n_l, long n_u, double q_lo,
                       double q_hi, void* state, float* low, float* high, void* stream);
/*      the same in phases for data-parallel runs: all-reduce the field at arco_sel_state_offset(0) (uint64 valid count)
 *      after phase 0 and the field at arco_sel_state_offset(1) (uint32 hist[4][256]) after every phase 2; phases:
 *      0 clear + count, 1 ranks, 2 histogram of digit `pass`, 3 pick digit `pass`, 4 masks                              */
long arco_sel_state_offset(int field);
int arco_entropy_masks_phase(int phase, int pass, const float* ent, const int64_t* lab_l, const int64_t* lab_u, long n_l,
                             long n_u, double q_lo, double q_hi, void* state, float* low, float* high, void* stream);

/* ---- L4  HOST entry points (CPU memory, no stream): native replay of the stratified samplers
 *      grid_monte_carlo_sample / grid_as_monte_carlo_sample (loss_helper_3d.py:120-268) on the
 *      serialized torch CPU generator state (torch.get_rng_state()), bit-exact incl. final state.        */
long arco_grid_sample(uint8_t* state, long state_bytes, long high, long shape, int cut, int mirror, int64_t* out);
/* a sequence of sampler calls in generator order; value-independent calls run on state copies in worker threads
 * while the generator is skipped ahead.  Returns the index of the first call that needs the 1-D fallback
 * (n_jobs when all ran; `state` = generator state right before that call), <0 on error.                       */
long arco_grid_sample_many(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                           int mirror, int64_t* const* outs, int max_threads);
/* deferred form: returns once the generator's final state is known (inline calls done, worker calls launched - their
   draw counts are fixed); arco_grid_sample_many_finish() waits for the worker calls' outputs.  The trainer draws the
   equivariance warp (train_arco_2d.py:412, the next consumer of the CPU generator) and queues that pass in between.  */
long arco_grid_sample_many_async(uint8_t* state, long state_bytes, int n_jobs, const long* highs, const long* shapes, int cut,
                                 int mirror, int64_t* const* outs, int max_threads);
void arco_grid_sample_many_finish(void);
long arco_randint(uint8_t* state, long state_bytes, long high, long n, int64_t* out);
/* generator consumption of a draw whose values are not needed: advance the serialized torch CPU generator by n 32-bit
 * draws (train_arco_2d.py:156 `torch.randn(K, 496, H, W)` when the revisiting term is off)                              */
long arco_mt_skip(uint8_t* state, long state_bytes, uint64_t n);
/* the generator's next state blocks for >= n_draws draws, computed ahead of time from `state` (not modified; in a worker
   thread when background != 0) while the host waits for the GPU's per-class counters (loss_helper_3d.py:413-434 needs them
   before the first sampler call :435-476).  The next arco_grid_sample_many call that starts from exactly this state reads
   the blocks instead of regenerating (skip-ahead O(1): the big calls run in parallel, the 16 grid blocks of an anchor call
   too); from any other state they are ignored.  Same draws, same final generator state.                              */
long arco_mt_pregen(const uint8_t* state, long state_bytes, long n_draws, int background);

/* ---- f16 ACTIVATION STORAGE of the volume path (BASELINE.json configs[4] "fp16 MFMA conv"; csrc/conv_h.hip) --------------
 * The V-Net body (vnetWithArgs.py:5-31 ConvBlock, :67-91 DownsamplingConvBlock, :94-118 UpsamplingDeconvBlock, :145-252 VNet)
 * with every activation and activation gradient held as f16 in HBM; weights, BatchNorm statistics, parameter gradients,
 * loss and optimizer fp32.  Entry points that take f16 tensors:
 *   arco_conv3d_fwd(..., mma = 4):   `in` and `out` are f16 rows (taps 27 with K = 1: `in` is the fp32 one-channel volume, `out`
 *                                    f16), Wp is the f16 pack of arco_pack_conv_weight(mode | 4) = [taps][ceil16(N)][ceil32(K)]
 *                                    halves (the K = 1 layer: the fp32 pack); v_mfma_f32_16x16x32_f16, fp32 accumulate, BN partials
 *                                    of the rounded outputs; no residual operand.
 *   arco_conv3d_wgrad(..., mma = 4): dZ and `in` f16 (Cin = 1: `in` fp32), dW fp32; operands transposed by ds_read_b64_tr_b16.
 *   arco_chan_stats_h / arco_bn_act_fwd_h / arco_bn_act_bwd_h / arco_colsum_h: the fp32 entry points of the same name on f16
 *                                    tensors (fp32 arithmetic, statistics and parameter gradients).
 *   arco_cast_h2f / arco_cast_f2h:   the region's boundary: outputs to fp32; fp32 gradients in, multiplied by the loss scale.
 * Space-to-depth (arco_s2d3) is a permutation of 4-byte words: f16 tensors pass with C / 2 and ld / 2.                       */
int arco_chan_stats_h(const void* X, long ldx, long M, int C, float* ssum, float* ssq, int groups, void* stream);
int arco_bn_act_fwd_h(const void* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                      const float* beta, float slope, int drop_mode, float p, uint64_t seed, long P, void* A, long lda,
                      const uint64_t* seed_dev, int groups, void* stream);
int arco_bn_act_bwd_h(const void* dA, long ldd, const void* Z, long ldz, long M, int C, const float* mean,
                      const float* istd, const float* gamma, const float* beta, float slope, int drop_mode, float p,
                      uint64_t seed, long P, float* ws, float* dgamma, float* dbeta, int accumulate, void* dZ, long ldo,
                      const uint64_t* seed_dev, int groups, void* stream);
int arco_colsum_h(const void* X, long ldx, long M, int C, float* ws, float* out, int accumulate, void* stream);
/* A = lrelu(BN(Z)) + R: the apply pass of UpsamplingDeconvBlock's BatchNorm with the decoder's skip addition folded in
 * (vnetWithArgs.py:224-236 `x5_up = self.block_five_up(x5) + x4` ...); mean / istd rows [groups][C] as for arco_bn_act_fwd      */
int arco_bn_act_add_fwd(const float* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                        const float* beta, float slope, const float* R, long ldr, float* A, long lda, int groups, void* stream);
int arco_bn_act_add_fwd_h(const void* Z, long ldz, long M, int C, const float* mean, const float* istd, const float* gamma,
                          const float* beta, float slope, const void* R, long ldr, void* A, long lda, int groups, void* stream);
/* UpsamplingDeconvBlock's BatchNorm + ReLU (+ skip) straight from the GEMM form of the k2 s2 transposed conv (vnetWithArgs.py:94-118):
 * Y = [voxels of the NV x X2 x Y2 x Z2 grid][8 C] (tap-major channels, tap = dx*4 + dy*2 + dz) is read as M8 = 8 * voxels rows of C
 * channels - the pre-activation in (voxel, tap) row order; statistics: arco_chan_stats(Y, ld = C, M8, C); the apply pass writes row
 * (voxel, tap) to voxel (n, 2x+dx, 2y+dy, 2z+dz) of A (and adds R there), the backward reads dA from there and leaves dY in Y's
 * order - no depth-to-space / space-to-depth pass in either direction.                                                            */
int arco_bn_act_d2s_fwd(const float* Y, long M8, int C, const float* mean, const float* istd, const float* gamma, const float* beta,
                        float slope, const float* R, long ldr, float* A, long lda, int X2, int Y2, int Z2, int groups, void* stream);
int arco_bn_act_d2s_fwd_h(const void* Y, long M8, int C, const float* mean, const float* istd, const float* gamma, const float* beta,
                          float slope, const void* R, long ldr, void* A, long lda, int X2, int Y2, int Z2, int groups, void* stream);
int arco_bn_act_d2s_bwd(const float* dA, long ldd, const float* Y, long M8, int C, const float* mean, const float* istd,
                        const float* gamma, const float* beta, float slope, float* ws, float* dgamma, float* dbeta, int accumulate,
                        float* dY, int X2, int Y2, int Z2, int groups, void* stream);
int arco_bn_act_d2s_bwd_h(const void* dA, long ldd, const void* Y, long M8, int C, const float* mean, const float* istd,
                          const float* gamma, const float* beta, float slope, float* ws, float* dgamma, float* dbeta, int accumulate,
                          void* dY, int X2, int Y2, int Z2, int groups, void* stream);
/* V = depth_to_space(P) + ADD: the gradient of an encoder activation that feeds both the next DownsamplingConvBlock (through
 * space-to-depth) and the decoder's skip connection (vnetWithArgs.py:186-201,224-236), one pass instead of arco_s2d3 + an add    */
int arco_d2s3_add(const float* P, long ldp, int NV, int X2, int Y2, int Z2, int C, const float* ADD, long lda, float* V, long ldv, void* stream);
int arco_d2s3_add_h(const void* P, long ldp, int NV, int X2, int Y2, int Z2, int C, const void* ADD, long lda, void* V, long ldv, void* stream);
int arco_cast_h2f(const void* x, long n, float* y, void* stream);
int arco_cast_f2h(const float* x, long n, float scale, void* y, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARCO_HIP_H */
