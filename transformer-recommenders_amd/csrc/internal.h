// Entry points shared between the translation units of libxfmr_hip.so but NOT part of the public C ABI
// (include/xfmr_hip.h): the same operators with type-erased activation pointers and a storage mask, used by the
// whole-encoder launch sequences to keep MFMA-only intermediates in HBM as bf16 (see common.h "Activation
// storage"). The public functions are these with mask 0.
#pragma once
#include "common.h"

enum : uint32_t {
  XF_S16_A = 1u,    // GEMM operand A (activations / incoming gradient)
  XF_S16_B = 2u,    // GEMM operand B when it is an activation (dW)
  XF_S16_C = 4u,    // GEMM output (and the pre-activation side output of the GELU epilogue)
  XF_S16_P = 8u,    // pre-activation input of the gelu' epilogue
  // (not a storage bit) the GELU Linear's auxiliary tensor holds gelu'(pre) instead of pre: the forward epilogue has
  // erf and exp(-x^2/2) in registers anyway, and the backward dX epilogue becomes one multiply
  XF_AUX_GELU_GRAD = 0x100u,
};

extern "C" {
int xf_linear_fwd_ex(const void* x, const float* w, const float* bias, void* y, int64_t M, int32_t N, int32_t K,
                     int32_t epilogue, const float* residual, void* aux_out, float dropout_p, XfSeed seed,
                     uint32_t site, int32_t precision, uint32_t s16, hipStream_t st);
// FFN forward in one kernel (bf16 storage, H = 128, I a multiple of 128): u16 <- bf16(x W1^T + b1), g16 <- gelu(u16) (either may be null),
// pre <- dropout(g16 W2^T + b2) + residual, y / y16 / mean / rstd <- LayerNorm(pre). gemm.hip: ffn_fwd_fused_kernel.
int xf_ffn_fwd_fused_ex(const void* x16, const void* w1_16, const float* b1, const void* w2_16, const float* b2,
                        void* u16, void* g16, float* pre, int64_t M, int32_t H, int32_t I, const float* residual, float dropout_p,
                        XfSeed seed, uint32_t site, const float* gamma, const float* beta, float eps, float* y,
                        void* y16, float* mean, float* rstd, hipStream_t st);
// The dX chain of the FFN backward in one kernel (bf16 storage, H = 128, I a multiple of 64; after the fused forward:
// u16 = the saved pre-activation): di16 <- (dy16 W2) * gelu'(u16), then exactly xf_linear_bwd_dx_lnbwd_ex on di16 / W1.
int xf_ffn_bwd_dx_fused_ex(const void* dy16, const void* w2_16, const void* u16, const void* w1_16, void* di16, int64_t M,
                           int32_t H, int32_t I, const float* residual_grad, const float* ln_x, const float* ln_mean,
                           const float* ln_rstd, const float* ln_gamma, float dropout_p, XfSeed seed, uint32_t site,
                           float* dx, void* d_lin16, float* partials, int* blocks_out, hipStream_t st);
int xf_linear_bwd_dx_ex(const void* dy, const float* w, void* dx, int64_t M, int32_t N, int32_t K,
                        const float* residual_grad, const void* gelu_pre, int32_t precision, uint32_t s16,
                        hipStream_t st);
int xf_linear_bwd_dw_ex(const void* dy, const void* x, float* dw, int64_t M, int32_t N, int32_t K, int32_t precision,
                        void* workspace, size_t workspace_bytes, uint32_t s16, hipStream_t st);
int xf_colsum_ex(const void* a, bool a16, float* out, int64_t M, int32_t N, void* workspace, hipStream_t st);
// Deferred form of the weight gradient for the whole-encoder backward: the split-K slabs [splits][N*K] are left in
// `slabs` and, when bias_part is given, the column sums of dy (the Linear's bias gradient) are left as partial rows
// [splits][N] -- both are reduced later by ONE xf_multi_rowsum launch for the whole backward instead of one
// (dW) + two (bias) small launches per Linear. Returns the number of splits in *splits.
size_t xf_linear_bwd_dw_slab_bytes(int64_t M, int32_t N, int32_t K);
int xf_linear_bwd_dw_deferred(const void* dy, const void* x, int64_t M, int32_t N, int32_t K, int32_t precision,
                              uint32_t s16, float* slabs, float* bias_part, int* splits, hipStream_t st);
// up to four of them over the same M tokens in one launch (gemm.hip: gemm_group_kernel)
struct XfDwItem { const void* dy; const void* x; int32_t N, K; float* slabs; float* bias_part; int* splits; };
int xf_linear_bwd_dw_group(const XfDwItem* items, int n, int64_t M, int32_t precision, uint32_t s16, hipStream_t st);
// the slab plan of one weight-gradient GEMM over M tokens (gemm.hip): token chunk per slab, returns the slab count
int xf_dw_split_plan(int64_t M, int32_t N, int32_t K, int* k_chunk);
// dw_ring.hip: the LDS-DMA ring form of the same GEMMs (same plan, same slab layout) for the shapes it takes
bool xf_dw_ring_takes(const XfDwItem* items, int n, int64_t M, int32_t precision, uint32_t s16);
int xf_dw_ring_launch(const XfDwItem* items, int n, int64_t M, hipStream_t st);
// dst[c] = sum_r src[r * ld + c], r < rows, c < cols, for every segment, in one launch (deterministic order)
struct XfReduceSeg { const float* src; float* dst; int rows; int cols; int ld; int pad; };
int xf_multi_rowsum(const XfReduceSeg* segs, int nseg, hipStream_t st);
int xf_rowsum(float* dst, const float* src, int64_t rows, int64_t cols, hipStream_t st);
int xf_attn_fwd_ex(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int32_t B, int32_t L, int32_t A,
                   int32_t H, float dropout_p, XfSeed seed, uint32_t site, int32_t precision, bool s16,
                   bool causal, hipStream_t st, const int32_t* seq_offsets = nullptr);  // packed rows: xfmr_encoder_cfg
int xf_attn_bwd_ex(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse, const void* d_ctx,
                   void* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p, XfSeed seed,
                   uint32_t site, int32_t precision, bool s16, bool causal, hipStream_t st,
                   const int32_t* seq_offsets = nullptr);
// loss.hip: deterministic totals of per-block fp64 records [24][nblocks] (each block = rows_per_block queries) ->
// losses[14], stats[16]; counts = device {n_valid, n_query}; tot = 24 doubles of scratch
int xf_loss_finalize(const double* blockpart, int nblocks, int rows_per_block, const int* counts, int mode,
                     int64_t n_rows, int64_t positions, float* losses, float* stats, double* tot, hipStream_t st);
// d_lin (optional): the gradient of the Linear output feeding this LayerNorm (dropout-scaled dx), bf16 if lin16
// d_gamma == d_beta == d_bias == nullptr defers the reduction of `partials` ([blocks][3][H], blocks returned in
// *blocks_out) to the caller (xf_multi_rowsum)
// Linear (+ bias, dropout, residual) with the LayerNorm that follows it applied in the GEMM epilogue (N == 128, bf16
// policy): pre = the LayerNorm input (saved for the backward), y / y16 = the output, mean / rstd per row.
int xf_linear_ln_fwd_ex(const void* x, const float* w, const float* bias, float* pre, int64_t M, int32_t N, int32_t K,
                        const float* residual, float dropout_p, XfSeed seed, uint32_t site, const float* gamma,
                        const float* beta, float eps, float* y, void* y16, float* mean, float* rstd, int32_t precision,
                        uint32_t s16, hipStream_t st);
// dx_ln_in[M,128] = LayerNormBackward(dy[M,N] * w[N,128] + residual_grad) in one kernel (bf16 policy): the dX GEMM whose
// output is the gradient of a LayerNorm output applies that LayerNorm's backward in its epilogue. d_lin16 (bf16,
// optional) = the dropout-scaled copy; partials[*blocks_out][3][128] = d gamma / d beta / d bias partial records.
// out_dropout_p / out_site: dropout that was applied to the LayerNorm OUTPUT (the embedding LayerNorm).
// number of 64-row M-tiles (one LayerNorm partial record each) the whole-row kernels run for M rows: >= (M + 63) / 64
// (short last-round tiles: gemm.hip xf_plan_row_tiles)
int xf_ln_row_tiles(int64_t M);
int xf_linear_bwd_dx_lnbwd_ex(const void* dy, const float* w, int64_t M, int32_t N, int32_t K,
                              const float* residual_grad, const float* ln_x, const float* ln_mean, const float* ln_rstd,
                              const float* ln_gamma, float dropout_p, XfSeed seed, uint32_t site, float* dx,
                              void* d_lin16, float* partials, int* blocks_out, int32_t precision, uint32_t s16,
                              hipStream_t st, float out_dropout_p = 0.f, uint32_t out_site = 0);
// LayerNorm forward variants that also write a bf16 copy of the output (the operand of the GEMMs that consume it;
// the fp32 output stays the residual stream). y16 / out16 may be null.
int xf_layernorm_fwd_ex(const float* x, const float* gamma, const float* beta, float* y, void* y16, float* mean,
                        float* rstd, int64_t rows, int32_t H, float eps, hipStream_t stream);
int xf_embed_ln_fwd_ex(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                       const float* type_emb, const float* gamma, const float* beta, float* out, void* out16,
                       float* pre, float* mean, float* rstd, uint8_t* key_mask, int32_t B, int32_t L, int32_t H,
                       float eps, float dropout_p, XfSeed seed, uint32_t site, hipStream_t stream);
int xf_embed_ln_fwd_packed_ex(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                              const float* type_emb, const float* gamma, const float* beta, float* out, void* out16,
                              float* pre, float* mean, float* rstd, uint8_t* key_mask, int64_t rows, const int32_t* row_pos,
                              int32_t H, float eps, float dropout_p, XfSeed seed, uint32_t site, hipStream_t stream);
int xf_embed_param_grads_packed(const float* d_pre, float* d_pos, float* d_type, const int32_t* offs, int32_t B, int32_t L,
                                int32_t H, int32_t max_pos, hipStream_t st);
int xf_layernorm_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          float* dx, void* d_lin, bool lin16, float* d_gamma, float* d_beta, float* d_bias,
                          int64_t rows, int32_t H, XfDropout drop_out, XfDropout drop_lin, void* partials,
                          hipStream_t st, int* blocks_out = nullptr);
}
