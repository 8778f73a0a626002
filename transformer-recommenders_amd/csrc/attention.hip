// Causal self-attention with key-padding mask, flash-style (scores never reach HBM): head size 32 (production kernels,
// every BASELINE config and the reference's default 384 / 12) or 64 (generic kernels).
//
// Layout trick used by all three kernels (forward, dQ, dK/dV): every score tile is produced with the
// softmax-reduction axis in the accumulator REGISTERS and the other axis on the LANE
// (forward/dQ: S^T = K Q^T, key in registers, query on the lane; dK/dV: S = Q K^T, query in registers,
// key on the lane). Row statistics are then lane-local, and the probability tile feeds the next MFMA as
// its B operand straight from the accumulator registers (no LDS round trip, no lane shuffles):
//   O^T  += V^T  P^T      dQ^T += K^T dS^T      dV^T += dO^T (P.D)      dK^T += Q^T dS
// One workgroup = 4 waves = 128 consecutive rows (queries, or keys for dK/dV) of one (batch, head);
// the whole K/V (or Q/dO) panel of that head that the causal mask can reach is staged once into LDS
// (L <= 512 keeps it under 160 KiB), so there is a single barrier per kernel.
#include "internal.h"

namespace {

constexpr int DH = 32;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct AttnArgs {
  const float* qkv; const uint8_t* key_mask; float* ctx; float* lse;  // qkv / ctx / d_ctx / d_qkv: fp32, or bf16
  const float* d_ctx; float* d_qkv;                                   // behind the same pointers (S16 kernels)
  int B, L, A, H;
  const int32_t* offs;  // packed rows (xfmr_encoder_cfg.seq_offsets): sequence b = rows [offs[b], offs[b + 1]) of qkv / ctx /
                        // key_mask / d_ctx / d_qkv, at most L of them; null: the padded layout, rows [b L, (b + 1) L). The
                        // one-workgroup-per-(batch, head) kernels only. lse and the dropout row keys stay indexed by L.
  int causal;  // 1: key j is visible to query i only when j <= i (BertConfig.is_decoder=True, the reference's
               // setting, models.py:355); 0: every unpadded key is visible (is_decoder=False)
  XfDropout drop;
};

// img[r][0..DHT) = src[off + (row0 + r) * stride + 0..DHT) for r in [0, nrows); rows >= row_end read as zero.
// (generic kernels below: head size DHT = 32 or 64, operands fp32 or -- S16 -- bf16 in HBM)
template <class P, int DHT, bool S16>
__device__ __forceinline__ void stage_rows(typename P::elem* img, int ld, const void* src, int64_t off, int64_t stride,
                                           int row0, int nrows, int row_end) {
  constexpr int CPR = DHT / 4;  // 4-element pieces per row
  for (int c = threadIdx.x; c < nrows * CPR; c += blockDim.x) {
    const int r = c / CPR, dd = (c % CPR) * 4;
    float4 v = make_float4(0, 0, 0, 0);
    if (row0 + r < row_end) v = xf_ld4<S16>(src, off + (int64_t)(row0 + r) * stride + dd);
    xf_store4<P>(img + r * ld + dd, v);
  }
}
// imgT[d][r] = src[off + (row0 + r) * stride + d]
template <class P, int DHT, bool S16>
__device__ __forceinline__ void stage_rows_T(typename P::elem* imgT, int ldT, const void* src, int64_t off, int64_t stride,
                                             int row0, int nrows, int row_end) {
  constexpr int CPR = DHT / 4;
  for (int c = threadIdx.x; c < (nrows / 2) * CPR; c += blockDim.x) {
    const int rp = c / CPR, dd = (c % CPR) * 4;
    float4 v0 = make_float4(0, 0, 0, 0), v1 = v0;
    const int r = 2 * rp;
    if (row0 + r < row_end) v0 = xf_ld4<S16>(src, off + (int64_t)(row0 + r) * stride + dd);
    if (row0 + r + 1 < row_end) v1 = xf_ld4<S16>(src, off + (int64_t)(row0 + r + 1) * stride + dd);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf_store2<P>(imgT + (dd + j) * ldT + r, xf_get(v0, j), xf_get(v1, j));
  }
}
// this lane's row of a register-resident operand, from fp32 or bf16 storage
template <class P, int DHT, bool S16>
__device__ __forceinline__ void load_reg_rows(RegRows<P, DHT>& reg, const void* base, int64_t idx, bool valid) {
  if constexpr (S16) reg.load(reinterpret_cast<const __bf16*>(base) + idx, valid);
  else reg.load(reinterpret_cast<const float*>(base) + idx, valid);
}

template <class P, int DHT = DH>
struct AttnSmem {
  using elem = typename P::elem;
  static constexpr int LDR = xf_ld<P>(DHT);  // row images [row][DHT]
  // transposed images [DHT][rows]: rows + 4 keeps 8-byte row alignment and makes the b64 fragment reads
  // of 32 consecutive image rows hit 32 distinct even bank pairs (conflict-free)
  __host__ __device__ static int ldt(int rows) { return rows + 4; }
  __host__ __device__ static size_t row_img(int rows) { return (size_t)rows * LDR * sizeof(elem); }
  __host__ __device__ static size_t t_img(int rows) { return (size_t)DHT * ldt(rows) * sizeof(elem); }
  __host__ __device__ static size_t align(size_t x) { return (x + 15) & ~(size_t)15; }
};
template <int DHT>
__host__ __device__ constexpr float attn_scale() { return DHT == 64 ? 0.125f : 0.17677669529663687f; }  // 1 / sqrt(DHT)

// ------------------------------------------------------------------------------------------------ forward
// Generic kernels (both precision policies, head size 32 or 64, fp32 or bf16 operands in HBM). The bf16 policy at head
// size 32 -- the reference's 384 / 12 and every BASELINE config -- runs the production kernels further down; these
// serve the fp32 parity policy and head size 64 (e.g. 384 / 6, 768 / 12: models.py:22-48 takes any hidden_size /
// num_attention_heads pair).
template <class P, int DHT, bool S16>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  using elem = typename P::elem;
  using SM = AttnSmem<P, DHT>;
  constexpr int ND = DHT / 32;  // 32-wide blocks of the head dimension
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const int b = blockIdx.y / a.A, h = blockIdx.y % a.A;
  const int qblk0 = blockIdx.x * 128;
  // keys the mask can reach, padded to 32: the causal triangle stops at the block's last query
  const int nkeys = a.causal ? min(((L + 31) / 32) * 32, qblk0 + 128) : ((L + 31) / 32) * 32;
  elem* sK = reinterpret_cast<elem*>(smem_raw);
  elem* sVT = reinterpret_cast<elem*>(smem_raw + SM::align(SM::row_img(nkeys)));
  const int ldt = SM::ldt(nkeys);
  float* scratch = reinterpret_cast<float*>(smem_raw + SM::align(SM::row_img(nkeys)) + SM::align(SM::t_img(nkeys)));
  uint8_t* sMask = reinterpret_cast<uint8_t*>(scratch + 4 * 32 * 33);

  const int64_t tok0 = (int64_t)b * L;
  const int64_t koff = tok0 * 3 * H + H + h * DHT, voff = tok0 * 3 * H + 2 * H + h * DHT;
  stage_rows<P, DHT, S16>(sK, SM::LDR, a.qkv, koff, 3 * H, 0, nkeys, L);
  stage_rows_T<P, DHT, S16>(sVT, ldt, a.qkv, voff, 3 * H, 0, nkeys, L);
  for (int t = threadIdx.x; t < nkeys; t += blockDim.x) sMask[t] = (t < L) ? a.key_mask[tok0 + t] : 0;
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int q0 = qblk0 + wid * 32;
  if (q0 >= L) return;
  const int q = q0 + (lane & 31);
  RegRows<P, DHT> qreg;
  load_reg_rows<P, DHT, S16>(qreg, a.qkv, (tok0 + q) * 3 * H + h * DHT, q < L);

  const float sc = attn_scale<DHT>() * kLog2e;  // 1/sqrt(head size) in base-2 exponent units
  float m = -INFINITY, lsum = 0.f;
  f32x16 o[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
  const uint32_t rowkey = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blockIdx.y * L + q));

  const int kb_end = a.causal ? min((q0 + 31) / 32, nkeys / 32 - 1) : nkeys / 32 - 1;
  for (int kb = 0; kb <= kb_end; ++kb) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    P::tile_nreg(s, sK, SM::LDR, kb * 32, qreg.regs(), DHT);
    float bmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + xf_acc_row(r, lane);
      const bool vis = (key <= q || !a.causal) && sMask[key];
      s[r] = vis ? s[r] * sc : -INFINITY;
      bmax = fmaxf(bmax, s[r]);
    }
    bmax = fmaxf(bmax, xf_half_swap(bmax));
    const float mnew = fmaxf(m, bmax);
    if (__all(mnew == -INFINITY)) continue;  // nothing visible yet for any query of this wave
    const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
    const float alpha = exp2f(m - msafe);  // m = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = exp2f(s[r] - msafe);  // masked: exp2(-inf) = 0
      psum += p;
      s[r] = a.drop.on ? p * xf_keep_scale_rc(a.drop, rowkey, (uint32_t)(kb * 32 + xf_acc_row(r, lane)) * kDropColMul) : p;
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int j = 0; j < ND; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[j][r] *= alpha;
    m = mnew;
#pragma unroll
    for (int j = 0; j < ND; ++j) P::tile_xb(o[j], sVT, ldt, 32 * j, kb * 32, s);
  }
  const float ltot = lsum + xf_half_swap(lsum);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
#pragma unroll
  for (int j = 0; j < ND; ++j)
    xf_store_tile_T_at<S16>(scratch + wid * 32 * 33, o[j], inv, a.ctx, tok0 * H + h * DHT + 32 * j, H, q0, L);
  if (lane < 32 && q < L)
    a.lse[((int64_t)blockIdx.y) * L + q] = ltot > 0.f ? (m + log2f(ltot)) * kLn2 : INFINITY;
}

// ------------------------------------------------------------------------------------------------ dQ
template <class P, int DHT, bool S16>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  using elem = typename P::elem;
  using SM = AttnSmem<P, DHT>;
  constexpr int ND = DHT / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const int b = blockIdx.y / a.A, h = blockIdx.y % a.A;
  const int qblk0 = blockIdx.x * 128;
  const int nkeys = a.causal ? min(((L + 31) / 32) * 32, qblk0 + 128) : ((L + 31) / 32) * 32;
  elem* sK = reinterpret_cast<elem*>(smem_raw);
  elem* sV = reinterpret_cast<elem*>(smem_raw + SM::align(SM::row_img(nkeys)));
  elem* sKT = reinterpret_cast<elem*>(smem_raw + 2 * SM::align(SM::row_img(nkeys)));
  const int ldt = SM::ldt(nkeys);
  float* scratch = reinterpret_cast<float*>(smem_raw + 2 * SM::align(SM::row_img(nkeys)) + SM::align(SM::t_img(nkeys)));
  uint8_t* sMask = reinterpret_cast<uint8_t*>(scratch + 4 * 32 * 33);

  const int64_t tok0 = (int64_t)b * L;
  const int64_t koff = tok0 * 3 * H + H + h * DHT, voff = tok0 * 3 * H + 2 * H + h * DHT;
  stage_rows<P, DHT, S16>(sK, SM::LDR, a.qkv, koff, 3 * H, 0, nkeys, L);
  stage_rows<P, DHT, S16>(sV, SM::LDR, a.qkv, voff, 3 * H, 0, nkeys, L);
  stage_rows_T<P, DHT, S16>(sKT, ldt, a.qkv, koff, 3 * H, 0, nkeys, L);
  for (int t = threadIdx.x; t < nkeys; t += blockDim.x) sMask[t] = (t < L) ? a.key_mask[tok0 + t] : 0;
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int q0 = qblk0 + wid * 32;
  if (q0 >= L) return;
  const int q = q0 + (lane & 31);
  const bool qv = q < L;
  RegRows<P, DHT> qreg, doreg;
  load_reg_rows<P, DHT, S16>(qreg, a.qkv, (tok0 + q) * 3 * H + h * DHT, qv);
  load_reg_rows<P, DHT, S16>(doreg, a.d_ctx, (tok0 + q) * H + h * DHT, qv);
  // delta = rowsum(dO * O) in fp32 from the stored tensors; each lane half covers half of the head's dims
  float delta = 0.f;
  if (qv) {
    const int64_t o0 = (tok0 + q) * H + h * DHT + (DHT / 2) * (lane >> 5);
#pragma unroll
    for (int u = 0; u < DHT / 8; ++u) {
      const float4 x = xf_ld4<S16>(a.ctx, o0 + 4 * u);
      const float4 y = xf_ld4<S16>(a.d_ctx, o0 + 4 * u);
      delta += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
  }
  delta += xf_half_swap(delta);
  const float lse2 = qv ? a.lse[(int64_t)blockIdx.y * L + q] * kLog2e : INFINITY;
  const float sc = attn_scale<DHT>() * kLog2e;
  const uint32_t rowkey = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blockIdx.y * L + q));

  f32x16 dq[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[j][r] = 0.f;
  const int kb_end = a.causal ? min((q0 + 31) / 32, nkeys / 32 - 1) : nkeys / 32 - 1;
  for (int kb = 0; kb <= kb_end; ++kb) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    P::tile_nreg(s, sK, SM::LDR, kb * 32, qreg.regs(), DHT);
    P::tile_nreg(dp, sV, SM::LDR, kb * 32, doreg.regs(), DHT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kb * 32 + xf_acc_row(r, lane);
      const bool vis = (key <= q || !a.causal) && sMask[key];
      const float p = vis ? exp2f(s[r] * sc - lse2) : 0.f;
      float dpv = dp[r];
      if (a.drop.on) dpv *= xf_keep_scale_rc(a.drop, rowkey, (uint32_t)key * kDropColMul);
      s[r] = p * (dpv - delta);
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) P::tile_xb(dq[j], sKT, ldt, 32 * j, kb * 32, s);
  }
#pragma unroll
  for (int j = 0; j < ND; ++j)
    xf_store_tile_T_at<S16>(scratch + wid * 32 * 33, dq[j], attn_scale<DHT>(), a.d_qkv, tok0 * 3 * H + h * DHT + 32 * j, 3 * H,
                            q0, L);
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <class P, int DHT, bool S16>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  using elem = typename P::elem;
  using SM = AttnSmem<P, DHT>;
  constexpr int ND = DHT / 32, CPR = DHT / 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const int b = blockIdx.y / a.A, h = blockIdx.y % a.A;
  const int kblk0 = blockIdx.x * 128;          // first key of this workgroup
  const int qs = a.causal ? kblk0 : 0;         // first query its keys are visible to
  const int Lp = ((L + 31) / 32) * 32;
  const int nq = Lp - qs;                       // queries [qs, Lp) staged, image row = q - qs
  elem* sQ = reinterpret_cast<elem*>(smem_raw);
  elem* sDO = reinterpret_cast<elem*>(smem_raw + SM::align(SM::row_img(nq)));
  elem* sQT = reinterpret_cast<elem*>(smem_raw + 2 * SM::align(SM::row_img(nq)));
  elem* sDOT = reinterpret_cast<elem*>(smem_raw + 2 * SM::align(SM::row_img(nq)) + SM::align(SM::t_img(nq)));
  const int ldt = SM::ldt(nq);
  float* scratch = reinterpret_cast<float*>(smem_raw + 2 * SM::align(SM::row_img(nq)) + 2 * SM::align(SM::t_img(nq)));
  float* sLse = scratch + 4 * 32 * 33;  // [nq] log2-scaled lse
  float* sDelta = sLse + nq;            // [nq]

  const int64_t tok0 = (int64_t)b * L;
  const int64_t qoff = tok0 * 3 * H + h * DHT, dooff = tok0 * H + h * DHT;
  stage_rows<P, DHT, S16>(sQ, SM::LDR, a.qkv, qoff, 3 * H, qs, nq, L);
  stage_rows<P, DHT, S16>(sDO, SM::LDR, a.d_ctx, dooff, H, qs, nq, L);
  stage_rows_T<P, DHT, S16>(sQT, ldt, a.qkv, qoff, 3 * H, qs, nq, L);
  stage_rows_T<P, DHT, S16>(sDOT, ldt, a.d_ctx, dooff, H, qs, nq, L);
  for (int c = threadIdx.x; c < nq * CPR; c += blockDim.x) {  // delta: CPR consecutive lanes share a row
    const int r = c / CPR, dd = (c % CPR) * 4;
    float part = 0.f;
    if (qs + r < L) {
      const float4 x = xf_ld4<S16>(a.ctx, dooff + (int64_t)(qs + r) * H + dd);
      const float4 y = xf_ld4<S16>(a.d_ctx, dooff + (int64_t)(qs + r) * H + dd);
      part = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
#pragma unroll
    for (int o = 1; o < CPR; o <<= 1) part += __shfl_xor(part, o, 64);
    if ((c % CPR) == 0) {
      sDelta[r] = part;
      sLse[r] = (qs + r < L) ? a.lse[(int64_t)blockIdx.y * L + qs + r] * kLog2e : INFINITY;
    }
  }
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int k0 = kblk0 + wid * 32;
  if (k0 >= L) return;
  const int key = k0 + (lane & 31);
  const bool kvis = key < L && a.key_mask[tok0 + (key < L ? key : 0)];
  RegRows<P, DHT> kreg, vreg;
  load_reg_rows<P, DHT, S16>(kreg, a.qkv, (tok0 + key) * 3 * H + H + h * DHT, key < L);
  load_reg_rows<P, DHT, S16>(vreg, a.qkv, (tok0 + key) * 3 * H + 2 * H + h * DHT, key < L);
  const float sc = attn_scale<DHT>() * kLog2e;

  f32x16 dk[ND], dv[ND];
#pragma unroll
  for (int j = 0; j < ND; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[j][r] = 0.f; dv[j][r] = 0.f; }
  for (int qb = a.causal ? k0 / 32 : 0; qb < Lp / 32; ++qb) {
    const int row0 = qb * 32 - qs;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    P::tile_nreg(s, sQ, SM::LDR, row0, kreg.regs(), DHT);
    P::tile_nreg(dp, sDO, SM::LDR, row0, vreg.regs(), DHT);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = row0 + xf_acc_row(r, lane);
      const int q = qi + qs;
      const bool vis = kvis && (key <= q || !a.causal);
      const float p = vis ? exp2f(s[r] * sc - sLse[qi]) : 0.f;  // q >= L: lse = +inf -> 0
      float keep = 1.f;
      if (a.drop.on) keep = xf_keep_scale_rc(a.drop, xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blockIdx.y * L + q)), (uint32_t)key * kDropColMul);
      s[r] = p * (dp[r] * keep - sDelta[qi]);  // dS
      dp[r] = p * keep;                        // P.D
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      P::tile_xb(dv[j], sDOT, ldt, 32 * j, row0, dp);
      P::tile_xb(dk[j], sQT, ldt, 32 * j, row0, s);
    }
  }
  float* sc_w = scratch + wid * 32 * 33;
#pragma unroll
  for (int j = 0; j < ND; ++j) {
    xf_store_tile_T_at<S16>(sc_w, dk[j], attn_scale<DHT>(), a.d_qkv, tok0 * 3 * H + H + h * DHT + 32 * j, 3 * H, k0, L);
    xf_store_tile_T_at<S16>(sc_w, dv[j], 1.f, a.d_qkv, tok0 * 3 * H + 2 * H + h * DHT + 32 * j, 3 * H, k0, L);
  }
}

// ================================================================================================================
// bf16 production kernels: ONE swizzled row image per operand (SwzImg<32>: unpadded 64-byte rows, chunk position
// c ^ ((row>>2)&3)); row-operand fragments are conflict-free 16-byte reads and the transposed fragments of the
// O / dQ / dK / dV products come from the SAME image through ds_read_b64_tr_b16 -- no transposed copies (V^T, K^T,
// Q^T, dO^T) and no scattered transposing stores while staging.
// ================================================================================================================
using AI = SwzImg<DH>;
// XCD-aware workgroup order (cf. gemm.hip tile_of): the 128-row blocks and the heads of ONE sequence read the same
// qkv / ctx / d_ctx rows (a 128-byte line holds two heads' slices), so all of a sequence's workgroups run on one XCD
// (d % 8) and share its L2. bx = 128-row block, by = b * A + h, as the (nblk, B*A) grid had them.
struct AttnBlock { int bx, by; bool valid; };
__device__ __forceinline__ AttnBlock attn_block(const AttnArgs& a) {
  const int nblk = (a.L + 127) / 128;
  const int d = blockIdx.x, xcd = d & 7, slot = d >> 3;
  const int per_b = nblk * a.A;
  const int b = (slot / per_b) * 8 + xcd, r = slot % per_b;
  return AttnBlock{r % nblk, b * a.A + r / nblk, b < a.B};
}
// bytes of the two staged row panels of a (batch, head), or of the 4 x 32 x 33 fp32 output-transposition scratch
// that later aliases them, whichever is larger (16-byte multiple)
__host__ __device__ inline size_t bf16_panel_bytes(int L) {
  const size_t Lp = ((size_t)L + 31) / 32 * 32;
  const size_t img = 2 * Lp * DH * 2, scr = 4 * 32 * 33 * sizeof(float);
  return ((img > scr ? img : scr) + 15) & ~(size_t)15;
}

// Stages TWO row panels (K and V, or Q and dO) into swizzled images. All of a thread's global loads of a batch
// (2 panels x 4 pieces) are issued before the first LDS store: the plain load -> store loop exposed one global
// latency per piece (4-7 round trips per workgroup before the first MFMA).
template <bool S16>
__device__ __forceinline__ void stage2_rows_swz(__bf16* imgA, const void* srcAv, int64_t offA, int64_t strideA,
                                                __bf16* imgB, const void* srcBv, int64_t offB, int64_t strideB,
                                                int row0, int nrows, int row_end) {
  constexpr int U = 4;
  if (S16) {  // bf16 storage: 16-byte chunks go to the image as they are (4 chunks per 32-wide row)
    const __bf16* srcA = reinterpret_cast<const __bf16*>(srcAv) + offA;
    const __bf16* srcB = reinterpret_cast<const __bf16*>(srcBv) + offB;
    const int total = nrows * 4;
    for (int c0 = threadIdx.x; c0 < total; c0 += (int)blockDim.x * U) {
      uint4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c >> 2, ch = c & 3;
        va[u] = vb[u] = make_uint4(0u, 0u, 0u, 0u);
        if (c < total && row0 + r < row_end) {
          va[u] = *reinterpret_cast<const uint4*>(srcA + (int64_t)(row0 + r) * strideA + ch * 8);
          vb[u] = *reinterpret_cast<const uint4*>(srcB + (int64_t)(row0 + r) * strideB + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c >> 2, ch = c & 3;
        if (c < total) {
          *reinterpret_cast<uint4*>(imgA + AI::off(r, ch)) = va[u];
          *reinterpret_cast<uint4*>(imgB + AI::off(r, ch)) = vb[u];
        }
      }
    }
  } else {
    const float* srcA = reinterpret_cast<const float*>(srcAv) + offA;
    const float* srcB = reinterpret_cast<const float*>(srcBv) + offB;
    const int total = nrows * 8;
    for (int c0 = threadIdx.x; c0 < total; c0 += (int)blockDim.x * U) {
      float4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c >> 3, dd = (c & 7) * 4;
        va[u] = vb[u] = make_float4(0, 0, 0, 0);
        if (c < total && row0 + r < row_end) {
          va[u] = *reinterpret_cast<const float4*>(srcA + (int64_t)(row0 + r) * strideA + dd);
          vb[u] = *reinterpret_cast<const float4*>(srcB + (int64_t)(row0 + r) * strideB + dd);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c >> 3, dd = (c & 7) * 4;
        if (c < total) {
          xf_store4<PrecBF16>(imgA + AI::off(r, dd >> 3) + (dd & 7), va[u]);
          xf_store4<PrecBF16>(imgB + AI::off(r, dd >> 3) + (dd & 7), vb[u]);
        }
      }
    }
  }
}
// sum over 16 consecutive elements of x[i] * y[i]
// key-padding mask of the (batch, head)'s keys as one bit per key (word kb = keys 32 kb .. 32 kb + 31): the score
// loops test a register bit instead of reading one LDS byte per score
__device__ __forceinline__ void stage_key_bits(uint32_t* bits, const uint8_t* key_mask, int nkeys, int L) {
  for (int t0 = 0; t0 < nkeys; t0 += (int)blockDim.x) {
    const int tt = t0 + (int)threadIdx.x;
    const bool f = tt < L && key_mask[tt] != 0;
    const unsigned long long bal = __ballot(f);
    if ((threadIdx.x & 63) == 0 && tt < nkeys) {
      bits[tt >> 5] = (uint32_t)bal;
      bits[(tt >> 5) + 1] = (uint32_t)(bal >> 32);  // (one word of slack is allocated)
    }
  }
}

template <bool S16>
__device__ __forceinline__ float dot16(const void* x, const void* y, int64_t off) {
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float4 a = xf_ld4<S16>(x, off + 4 * u), b = xf_ld4<S16>(y, off + 4 * u);
    acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  return acc;
}

template <bool S16>
__global__ __launch_bounds__(256, 4) void attn_fwd_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const AttnBlock blk = attn_block(a);
  if (!blk.valid) return;  // (whole workgroup: the grid is padded to a multiple of 8 sequences)
  const int b = blk.by / a.A, h = blk.by % a.A;
  const int qblk0 = blk.bx * 128;
  const int nkeys = a.causal ? min(((L + 31) / 32) * 32, qblk0 + 128) : ((L + 31) / 32) * 32;
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sV = sK + nkeys * DH;
  // the output-transposition scratch ALIASES the K / V images (used after a barrier, once every wave is done with
  // them): 46 KB -> 29 KB per workgroup at L = 200, so all 4 workgroups per CU of the (2, B x A) grid are resident
  // at once instead of 3 plus a one-third-full second round
  float* scratch = reinterpret_cast<float*>(smem_raw);
  uint32_t* sBits = reinterpret_cast<uint32_t*>(smem_raw + bf16_panel_bytes(L));  // key mask, one bit per key

  const int64_t tok0 = (int64_t)b * L;
  stage2_rows_swz<S16>(sK, a.qkv, tok0 * 3 * H + H + h * DH, 3 * H, sV, a.qkv, tok0 * 3 * H + 2 * H + h * DH, 3 * H, 0,
                       nkeys, L);
  stage_key_bits(sBits, a.key_mask + tok0, nkeys, L);
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int q0 = qblk0 + wid * 32;
  const bool active = q0 < L;  // (wave-uniform) inactive waves still take part in the barrier below
  const int q = q0 + (lane & 31);
  RegRows<PrecBF16, DH> qreg;
  qreg.load_at<S16>(a.qkv, (tok0 + q) * 3 * H + h * DH, active && q < L);
  const float sc = 0.17677669529663687f * kLog2e;
  float m = -INFINITY, lsum = 0.f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  const uint32_t rowkey = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * L + q));
  const int kb_end = !active ? -1 : a.causal ? min((q0 + 31) / 32, nkeys / 32 - 1) : nkeys / 32 - 1;
  for (int kb = 0; kb <= kb_end; ++kb) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    AI::tile_nreg(s, sK, kb * 32, qreg.regs());
    float bmax = -INFINITY;
    const uint32_t kword = sBits[kb];
    // interior tile: every key is at or before the wave's first query and none is padding -- no per-score masking
    const bool interior = (kb * 32 + 31 <= q0 || !a.causal) && __builtin_amdgcn_readfirstlane(kword) == 0xFFFFFFFFu;
    if (interior) {
#pragma unroll
      for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, s[r]);
      bmax *= sc;
    } else {
      const uint32_t kbits = kword >> (4 * (lane >> 5));  // this half-wave's keys: bit (r&3) + 8*(r>>2)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + xf_acc_row(r, lane);
        const bool vis = (key <= q || !a.causal) && ((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u);
        s[r] = vis ? s[r] : -INFINITY;
        bmax = fmaxf(bmax, s[r]);
      }
      bmax *= sc;  // (sc > 0; -inf stays -inf)
    }
    bmax = fmaxf(bmax, xf_half_swap(bmax));
    const float mnew = fmaxf(m, bmax);
    if (__all(mnew == -INFINITY)) continue;
    const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
    const float alpha = xf_exp2(m - msafe);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = xf_exp2(fmaf(s[r], sc, -msafe));
      psum += p;
      s[r] = a.drop.on ? p * xf_keep_scale_rc(a.drop, rowkey, (uint32_t)(kb * 32 + xf_acc_row(r, lane)) * kDropColMul) : p;
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    m = mnew;
    AI::tile_xb_tr(o, sV, 0, kb * 32, s);
  }
  const float ltot = lsum + xf_half_swap(lsum);
  const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
  __syncthreads();  // every wave is done with the K / V images the scratch aliases
  if (!active) return;
  xf_store_tile_T_at<S16>(scratch + wid * 32 * 33, o, inv, a.ctx, tok0 * H + h * DH, H, q0, L);
  if (lane < 32 && q < L)
    a.lse[((int64_t)blk.by) * L + q] = ltot > 0.f ? (m + log2f(ltot)) * kLn2 : INFINITY;
}

template <bool S16>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const AttnBlock blk = attn_block(a);
  if (!blk.valid) return;  // (whole workgroup: the grid is padded to a multiple of 8 sequences)
  const int b = blk.by / a.A, h = blk.by % a.A;
  const int qblk0 = blk.bx * 128;
  const int nkeys = a.causal ? min(((L + 31) / 32) * 32, qblk0 + 128) : ((L + 31) / 32) * 32;
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sV = sK + nkeys * DH;
  // the output-transposition scratch ALIASES the K / V images (used after a barrier, once every wave is done with
  // them): 46 KB -> 29 KB per workgroup at L = 200, so all 4 workgroups per CU of the (2, B x A) grid are resident
  // at once instead of 3 plus a one-third-full second round
  float* scratch = reinterpret_cast<float*>(smem_raw);
  uint32_t* sBits = reinterpret_cast<uint32_t*>(smem_raw + bf16_panel_bytes(L));  // key mask, one bit per key

  const int64_t tok0 = (int64_t)b * L;
  stage2_rows_swz<S16>(sK, a.qkv, tok0 * 3 * H + H + h * DH, 3 * H, sV, a.qkv, tok0 * 3 * H + 2 * H + h * DH, 3 * H, 0,
                       nkeys, L);
  stage_key_bits(sBits, a.key_mask + tok0, nkeys, L);
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int q0 = qblk0 + wid * 32;
  const bool active = q0 < L;
  const int q = q0 + (lane & 31);
  const bool qv = active && q < L;
  RegRows<PrecBF16, DH> qreg, doreg;
  qreg.load_at<S16>(a.qkv, (tok0 + q) * 3 * H + h * DH, qv);
  doreg.load_at<S16>(a.d_ctx, (tok0 + q) * H + h * DH, qv);
  float delta = 0.f;
  if (qv) delta = dot16<S16>(a.ctx, a.d_ctx, (tok0 + q) * H + h * DH + 16 * (lane >> 5));
  delta += xf_half_swap(delta);
  const float lse2 = qv ? a.lse[(int64_t)blk.by * L + q] * kLog2e : INFINITY;
  const float sc = 0.17677669529663687f * kLog2e;
  const uint32_t rowkey = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * L + q));
  f32x16 dq;
#pragma unroll
  for (int r = 0; r < 16; ++r) dq[r] = 0.f;
  const int kb_end = !active ? -1 : a.causal ? min((q0 + 31) / 32, nkeys / 32 - 1) : nkeys / 32 - 1;
  for (int kb = 0; kb <= kb_end; ++kb) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    AI::tile_nreg(s, sK, kb * 32, qreg.regs());
    AI::tile_nreg(dp, sV, kb * 32, doreg.regs());
    const uint32_t kword = sBits[kb];
    const bool interior = (kb * 32 + 31 <= q0 || !a.causal) && __builtin_amdgcn_readfirstlane(kword) == 0xFFFFFFFFu;
    if (interior) {  // (see attn_fwd_bf16_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = xf_exp2(fmaf(s[r], sc, -lse2));
        float dpv = dp[r];
        if (a.drop.on) dpv *= xf_keep_scale_rc(a.drop, rowkey, (uint32_t)(kb * 32 + xf_acc_row(r, lane)) * kDropColMul);
        s[r] = p * (dpv - delta);
      }
    } else {
      const uint32_t kbits = kword >> (4 * (lane >> 5));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb * 32 + xf_acc_row(r, lane);
        const bool vis = (key <= q || !a.causal) && ((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u);
        const float p = vis ? xf_exp2(fmaf(s[r], sc, -lse2)) : 0.f;
        float dpv = dp[r];
        if (a.drop.on) dpv *= xf_keep_scale_rc(a.drop, rowkey, (uint32_t)key * kDropColMul);
        s[r] = p * (dpv - delta);
      }
    }
    AI::tile_xb_tr(dq, sK, 0, kb * 32, s);
  }
  __syncthreads();  // the scratch aliases the K / V images
  if (!active) return;
  xf_store_tile_T_at<S16>(scratch + wid * 32 * 33, dq, 0.17677669529663687f, a.d_qkv, tok0 * 3 * H + h * DH, 3 * H, q0,
                          L);
}

template <bool S16>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int L = a.L, H = a.H;
  const AttnBlock blk = attn_block(a);
  if (!blk.valid) return;  // (whole workgroup: the grid is padded to a multiple of 8 sequences)
  const int b = blk.by / a.A, h = blk.by % a.A;
  const int kblk0 = blk.bx * 128;
  const int Lp = ((L + 31) / 32) * 32;
  const int qs = a.causal ? kblk0 : 0;  // first query this block's keys are visible to; image row = q - qs
  const int nq = Lp - qs;
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sDO = sQ + nq * DH;
  float* scratch = reinterpret_cast<float*>(smem_raw);  // aliases the Q / dO images: see attn_fwd_bf16_kernel
  float* sLse = reinterpret_cast<float*>(smem_raw + bf16_panel_bytes(L));
  float* sDelta = sLse + nq;
  uint32_t* sRowKey = reinterpret_cast<uint32_t*>(sDelta + nq);  // dropout row keys of the staged query rows

  const int64_t tok0 = (int64_t)b * L;
  const int64_t hoff = tok0 * H + h * DH;
  stage2_rows_swz<S16>(sQ, a.qkv, tok0 * 3 * H + h * DH, 3 * H, sDO, a.d_ctx, hoff, H, qs, nq, L);
  // delta[r] = rowsum(dO * O), lse[r]: loads of a batch of 4 pieces issued together (see stage2_rows_swz)
  for (int c0 = threadIdx.x; c0 < nq * 8; c0 += (int)blockDim.x * 4) {
    float4 x[4], y[4];
    float ls[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * (int)blockDim.x, r = c >> 3, dd = (c & 7) * 4;
      x[u] = y[u] = make_float4(0, 0, 0, 0);
      ls[u] = INFINITY;
      if (c < nq * 8 && qs + r < L) {
        x[u] = xf_ld4<S16>(a.ctx, hoff + (int64_t)(qs + r) * H + dd);
        y[u] = xf_ld4<S16>(a.d_ctx, hoff + (int64_t)(qs + r) * H + dd);
        if ((c & 7) == 0) ls[u] = a.lse[(int64_t)blk.by * L + qs + r] * kLog2e;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = c0 + u * (int)blockDim.x, r = c >> 3;
      float part = x[u].x * y[u].x + x[u].y * y[u].y + x[u].z * y[u].z + x[u].w * y[u].w;
      part += __shfl_xor(part, 1, 64);
      part += __shfl_xor(part, 2, 64);
      part += __shfl_xor(part, 4, 64);
      if (c < nq * 8 && (c & 7) == 0) {
        sDelta[r] = part;
        sLse[r] = ls[u];
        sRowKey[r] = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * L + qs + r));
      }
    }
  }
  __syncthreads();

  const int lane = xf_lane(), wid = threadIdx.x >> 6;
  const int k0 = kblk0 + wid * 32;
  const bool active = k0 < L;
  const int key = k0 + (lane & 31);
  const bool kvis = active && key < L && a.key_mask[tok0 + (key < L ? key : 0)];
  RegRows<PrecBF16, DH> kreg, vreg;
  kreg.load_at<S16>(a.qkv, (tok0 + key) * 3 * H + H + h * DH, active && key < L);
  vreg.load_at<S16>(a.qkv, (tok0 + key) * 3 * H + 2 * H + h * DH, active && key < L);
  const float sc = 0.17677669529663687f * kLog2e;
  f32x16 dk, dv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  const bool all_kvis = __all(kvis);
  const uint32_t colmix = (uint32_t)key * kDropColMul;
  for (int qb = !active ? Lp / 32 : a.causal ? k0 / 32 : 0; qb < Lp / 32; ++qb) {
    const int row0 = qb * 32 - qs;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    AI::tile_nreg(s, sQ, row0, kreg.regs());
    AI::tile_nreg(dp, sDO, row0, vreg.regs());
    const bool interior = all_kvis && (qb * 32 > k0 || !a.causal);  // every query row is after the wave's keys, every key valid
    // the lane's 16 query rows are four runs of 4 consecutive rows: lse / delta come as 16-byte LDS reads
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qi0 = row0 + 8 * g + 4 * (lane >> 5);
      const float4 l4 = *reinterpret_cast<const float4*>(&sLse[qi0]);
      const float4 d4 = *reinterpret_cast<const float4*>(&sDelta[qi0]);
      const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
      uint32_t rk[4] = {0u, 0u, 0u, 0u};
      if (a.drop.on) {
        const uint4 k4 = *reinterpret_cast<const uint4*>(&sRowKey[qi0]);
        rk[0] = k4.x; rk[1] = k4.y; rk[2] = k4.z; rk[3] = k4.w;
      }
      float pr[4];
      if (interior) {
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = xf_exp2(fmaf(s[4 * g + u], sc, -ls[u]));
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
          pr[u] = (kvis && (key <= qi0 + u + qs || !a.causal)) ? xf_exp2(fmaf(s[4 * g + u], sc, -ls[u])) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = 4 * g + u;
        const float p = pr[u];
        float keep = 1.f;
        if (a.drop.on) keep = xf_keep_scale_rc(a.drop, rk[u], colmix);
        s[r] = p * (dp[r] * keep - dl[u]);
        dp[r] = p * keep;
      }
    }
    AI::tile_xb_tr(dv, sDO, 0, row0, dp);
    AI::tile_xb_tr(dk, sQ, 0, row0, s);
  }
  __syncthreads();  // the scratch aliases the Q / dO images
  if (!active) return;
  float* sc_w = scratch + wid * 32 * 33;
  xf_store_tile_T_at<S16>(sc_w, dk, 0.17677669529663687f, a.d_qkv, tok0 * 3 * H + H + h * DH, 3 * H, k0, L);
  xf_store_tile_T_at<S16>(sc_w, dv, 1.f, a.d_qkv, tok0 * 3 * H + 2 * H + h * DH, 3 * H, k0, L);
}

// ---- fused backward: ONE workgroup per (batch, head) computes dQ, dK and dV -------------------------------------
// The two-kernel form stages Q / dO / ctx / d_ctx (dK/dV kernel) and K / V / q / dO / ctx / d_ctx (dQ kernel) per
// 128-row block and evaluates every probability twice; measured at B = 512, L = 200: 40 + 61 us of the 65 + 132 us
// were staging and stores alone, not overlapped with the score loops. Here Q and dO are staged once, a wave owns
// key tiles (dK, dV in registers, as before) and every score tile's dS additionally goes through a per-wave
// 32 x 32 bf16 image -- written from the accumulator layout with 8-byte stores, read back TRANSPOSED as the MFMA
// A operand -- for dQ[q][d] += sum_key dS[q][key] K[key][d], which is added into an fp32 [L][32] accumulator in LDS
// (ds_add_f32; the four waves reach a query tile at different times). Key tiles are dealt to the waves in snake
// order (w, 7-w, 8+w, 15-w, ...): tile t has (tiles - t) query tiles to walk, so the pairs balance.
// Lock-step schedule of the fused backward: kBwdSched[tiles - 1][wave][step] = key tile << 4 | query tile (255 =
// idle). A wave walks one key tile's query tiles contiguously (dK / dV stay in registers), the loads are balanced
// (snake deal of the key tiles) and within a step the four waves hold four DIFFERENT query tiles, so the dQ
// accumulator tile in LDS is read-modify-written without atomics (ds_add_f32 measured 2.5 cycles PER LANE: 244 us of
// a 354 us kernel) and in a fixed order: results are bit-reproducible. Found by exhaustive search over tile orders,
// rotations and directions (up to 8 tiles = L <= 256; longer sequences use the two-kernel form).
__constant__ uint8_t kBwdSched[8][4][9] = {
  {{0,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}},
  {{0,1,255,255,255,255,255,255,255}, {17,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}},
  {{0,1,2,255,255,255,255,255,255}, {17,18,255,255,255,255,255,255,255}, {34,255,255,255,255,255,255,255,255}, {255,255,255,255,255,255,255,255,255}},
  {{0,1,2,3,255,255,255,255,255}, {17,18,19,255,255,255,255,255,255}, {34,35,255,255,255,255,255,255,255}, {51,255,255,255,255,255,255,255,255}},
  {{0,1,2,3,4,255,255,255,255}, {19,18,17,20,255,255,255,255,255}, {34,35,36,255,255,255,255,255,255}, {68,52,51,255,255,255,255,255,255}},
  {{0,1,2,3,4,5,255,255,255}, {21,20,19,18,17,255,255,255,255}, {34,35,36,37,85,255,255,255,255}, {68,69,53,52,51,255,255,255,255}},
  {{0,1,2,3,4,5,6,255,255}, {102,22,21,20,19,18,17,255,255}, {34,35,36,37,38,86,85,255,255}, {68,69,70,54,53,52,51,255,255}},
  {{0,1,2,3,4,5,6,7,119}, {102,103,23,22,21,20,19,18,17}, {34,35,36,37,38,39,87,86,85}, {68,69,70,71,55,54,53,52,51}},
};
__constant__ uint8_t kBwdSteps[8] = {1, 2, 3, 4, 5, 6, 7, 9};
constexpr int kFusedMaxL = 256;  // the lock-step backward (its schedule table: <= 8 tiles)
constexpr int kSeqMaxL = 512;    // the one-workgroup forward and the two-role backward (<= 16 tiles: four per wave)

__device__ __forceinline__ AttnBlock attn_seq_block(const AttnArgs& a) {  // as attn_block with one block per (b, h)
  const int d = blockIdx.x, xcd = d & 7, slot = d >> 3;
  const int b = (slot / a.A) * 8 + xcd;
  return AttnBlock{0, b * a.A + slot % a.A, b < a.B};
}

// ---- forward, one workgroup per (batch, head) (L <= 256) -----------------------------------------------------------
// K and V are staged once for the whole sequence (the two-block form staged the first 128 keys twice) and the query
// tiles are dealt so that every wave walks the same number of key blocks: tile t has t + 1 of them, wave w takes tiles
// (tiles - 1 - w) and (tiles - 8 + w). The output goes straight from the accumulator layout (lane = query row) to
// global memory, so no wave waits for the others before it stores.
template <bool S16>
__global__ __launch_bounds__(256, 4) void attn_fwd_seq_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = a.H;
  int L = a.L;  // this sequence's rows (packed layout: its own length)
  const AttnBlock blk = attn_seq_block(a);
  if (!blk.valid) return;  // (whole workgroup)
  const int b = blk.by / a.A, h = blk.by % a.A;
  int64_t tok0 = (int64_t)b * a.L;
  if (a.offs) {  // packed rows: the sequence starts at its offset and is as long as it is
    const int o0 = a.offs[b];
    L = a.offs[b + 1] - o0;
    tok0 = o0;
    if (L <= 0) return;  // (whole workgroup)
  }
  const int Lp = ((L + 31) / 32) * 32, nt = Lp / 32;
  __bf16* sK = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sV = sK + Lp * DH;
  uint32_t* sBits = reinterpret_cast<uint32_t*>(sV + Lp * DH);  // key mask, one bit per key

  const int lane = xf_lane(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5;
  // the wave's query tiles (at most two up to 8 tiles, four up to 16): tile t walks t + 1 key blocks, the deal pairs long
  // with short (nt-1-w, nt-8+w | nt-9-w, nt-16+w). The first two tiles' query rows are in flight while K / V are staged.
  const int qt[4] = {nt - 1 - wid, nt - 8 + wid, nt - 9 - wid, nt - 16 + wid};
  const int ntile = nt > 8 ? 4 : 2;
  RegRows<PrecBF16, DH> qreg[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = qt[i] * 32 + (lane & 31);
    qreg[i].template load_at<S16>(a.qkv, (tok0 + q) * 3 * H + h * DH, qt[i] >= 0 && q < L);
  }
  stage2_rows_swz<S16>(sK, a.qkv, tok0 * 3 * H + H + h * DH, 3 * H, sV, a.qkv, tok0 * 3 * H + 2 * H + h * DH, 3 * H, 0,
                       Lp, L);
  stage_key_bits(sBits, a.key_mask + tok0, Lp, L);
  __syncthreads();

  const float sc = 0.17677669529663687f * kLog2e;
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // (unrolled: qt[i] / qreg[i & 1] are registers, not indexed arrays)
    if (i >= ntile || qt[i] < 0) continue;
    if (i >= 2) {  // (sequences longer than 256: the third / fourth tile's query rows are fetched when their turn comes)
      const int qq = qt[i] * 32 + (lane & 31);
      qreg[i & 1].template load_at<S16>(a.qkv, (tok0 + qq) * 3 * H + h * DH, qq < L);
    }
    const int q0 = qt[i] * 32, q = q0 + (lane & 31);
    float m = -INFINITY, lsum = 0.f;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const uint32_t rowkey = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * a.L + q));
    for (int kb = 0; kb <= qt[i]; ++kb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      AI::tile_nreg(s, sK, kb * 32, qreg[i & 1].regs());
      float bmax = -INFINITY;
      const uint32_t kword = sBits[kb];
      // interior tile: every key is before the wave's first query and none is padding -- no per-score masking
      const bool interior = kb < qt[i] && __builtin_amdgcn_readfirstlane(kword) == 0xFFFFFFFFu;
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bmax = fmaxf(bmax, s[r]);
        bmax *= sc;
      } else {
        const uint32_t kbits = kword >> (4 * hh);  // this half-wave's keys: bit (r&3) + 8*(r>>2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kb * 32 + xf_acc_row(r, lane);
          const bool vis = (key <= q || !a.causal) && ((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u);
          s[r] = vis ? s[r] : -INFINITY;
          bmax = fmaxf(bmax, s[r]);
        }
        bmax *= sc;  // (sc > 0; -inf stays -inf)
      }
      bmax = fmaxf(bmax, xf_half_swap(bmax));
      const float mnew = fmaxf(m, bmax);
      if (__all(mnew == -INFINITY)) continue;
      const float msafe = (mnew == -INFINITY) ? 0.f : mnew;
      const float alpha = xf_exp2(m - msafe);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = xf_exp2(fmaf(s[r], sc, -msafe));
        psum += p;
        s[r] = a.drop.on
                   ? p * xf_keep_scale_rc(a.drop, rowkey, (uint32_t)(kb * 32 + xf_acc_row(r, lane)) * kDropColMul)
                   : p;
      }
      lsum = lsum * alpha + psum;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= alpha;
      m = mnew;
      AI::tile_xb_tr(o, sV, 0, kb * 32, s);
    }
    const float ltot = lsum + xf_half_swap(lsum);
    const float inv = ltot > 0.f ? 1.f / ltot : 0.f;
    if (q < L) {  // o[r] <-> (d = acc_row(r), query = lane & 31): 4 consecutive d per register group
      const int64_t off = (tok0 + q) * H + h * DH;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        xf_st4<S16>(a.ctx, off + 8 * g + 4 * hh,
                    make_float4(o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv));
      if (lane < 32) a.lse[((int64_t)blk.by) * a.L + q] = ltot > 0.f ? (m + log2f(ltot)) * kLn2 : INFINITY;
    }
  }
}

size_t bf16_smem_fwd_seq(int L) {  // K + V images of the whole sequence, key mask bits
  const size_t Lp = ((size_t)L + 31) / 32 * 32;
  return 2 * Lp * DH * 2 + (Lp / 32 + 2) * sizeof(uint32_t);
}

template <bool S16>
__global__ __launch_bounds__(256, 2) void attn_bwd_fused_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = a.H;
  int L = a.L;  // this sequence's rows (packed layout: its own length)
  const AttnBlock blk = attn_seq_block(a);
  if (!blk.valid) return;  // (whole workgroup)
  const int b = blk.by / a.A, h = blk.by % a.A;
  int64_t tok0 = (int64_t)b * a.L;
  if (a.offs) {  // packed rows: the sequence starts at its offset and is as long as it is
    const int o0 = a.offs[b];
    L = a.offs[b + 1] - o0;
    tok0 = o0;
    if (L <= 0) return;  // (whole workgroup)
  }
  const int Lp = ((L + 31) / 32) * 32, nt = Lp / 32;
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sDO = sQ + Lp * DH;
  float* sDq = reinterpret_cast<float*>(sDO + Lp * DH);           // [Lp][32] fp32 dQ accumulator
  float* sLse = sDq + Lp * DH;
  float* sDelta = sLse + Lp;
  uint32_t* sRowKey = reinterpret_cast<uint32_t*>(sDelta + Lp);
  __bf16* sDS = reinterpret_cast<__bf16*>(sRowKey + Lp);          // [4 waves][32 keys][32 queries] swizzled

  const int64_t hoff = tok0 * H + h * DH;
  // Staging in ONE round trip: the Q, dO and ctx pieces of a row chunk are loaded together; Q and dO go to the
  // swizzled images, delta[r] = rowsum(dO * O) is reduced over the lanes that hold the row's chunks.
  for (int c = threadIdx.x; c < Lp * 8; c += (int)blockDim.x) reinterpret_cast<float4*>(sDq)[c] = make_float4(0, 0, 0, 0);
  {
    constexpr int U = 4, PPR = S16 ? 4 : 8;  // pieces per 32-wide row: 16-byte pieces of bf16 / fp32
    const int total = Lp * PPR;
    for (int c0 = threadIdx.x; c0 < total; c0 += (int)blockDim.x * U) {
      uint4 vq[U], vd[U], vo[U];
      float ls[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c / PPR, pc = c % PPR;
        vq[u] = vd[u] = vo[u] = make_uint4(0u, 0u, 0u, 0u);
        ls[u] = INFINITY;
        if (c < total && r < L) {
          const int e = pc * (32 / PPR);  // first element of the piece
          vq[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.qkv, (tok0 + r) * 3 * H + h * DH + e));
          vd[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.d_ctx, hoff + (int64_t)r * H + e));
          vo[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.ctx, hoff + (int64_t)r * H + e));
          if (pc == 0) ls[u] = a.lse[(int64_t)blk.by * a.L + r] * kLog2e;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c / PPR, pc = c % PPR;
        float part;
        if (S16) {
          if (c < total) {
            *reinterpret_cast<uint4*>(sQ + AI::off(r, pc)) = vq[u];
            *reinterpret_cast<uint4*>(sDO + AI::off(r, pc)) = vd[u];
          }
          const float4 d0 = xf_bf16x4_to_f32(make_uint2(vd[u].x, vd[u].y)), d1 = xf_bf16x4_to_f32(make_uint2(vd[u].z, vd[u].w));
          const float4 o0 = xf_bf16x4_to_f32(make_uint2(vo[u].x, vo[u].y)), o1 = xf_bf16x4_to_f32(make_uint2(vo[u].z, vo[u].w));
          part = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z +
                 d1.w * o1.w;
        } else {
          const float4 q4 = *reinterpret_cast<const float4*>(&vq[u]), d4 = *reinterpret_cast<const float4*>(&vd[u]),
                       o4 = *reinterpret_cast<const float4*>(&vo[u]);
          if (c < total) {
            const int dd = pc * 4;
            xf_store4<PrecBF16>(sQ + AI::off(r, dd >> 3) + (dd & 7), q4);
            xf_store4<PrecBF16>(sDO + AI::off(r, dd >> 3) + (dd & 7), d4);
          }
          part = d4.x * o4.x + d4.y * o4.y + d4.z * o4.z + d4.w * o4.w;
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if (!S16) part += __shfl_xor(part, 4, 64);
        if (c < total && pc == 0) {
          sDelta[r] = part;
          sLse[r] = ls[u];
          sRowKey[r] = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * a.L + r));
        }
      }
    }
  }

  const int lane = xf_lane(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5;
  __bf16* sDSw = sDS + wid * 32 * DH;
  const float sc = 0.17677669529663687f * kLog2e;
  // per-key-tile state of the wave
  int cur_kt = -1, key = 0;
  bool kin = false, kvis = false, all_kvis = false;
  uint32_t colmix = 0;
  f32x16 dk, dv;
  auto flush = [&]() {  // dK / dV straight from the accumulator layout: lane = key row, 4 consecutive d per group
    if (cur_kt >= 0 && kin) {
      const int64_t o = (tok0 + key) * 3 * H + h * DH;
      const float ks = 0.17677669529663687f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        xf_st4<S16>(a.d_qkv, o + H + 8 * g + 4 * hh,
                    make_float4(dk[4 * g] * ks, dk[4 * g + 1] * ks, dk[4 * g + 2] * ks, dk[4 * g + 3] * ks));
        xf_st4<S16>(a.d_qkv, o + 2 * H + 8 * g + 4 * hh,
                    make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]));
      }
    }
  };
  // a wave walks at most two key tiles (up to 8 tiles): the operands of BOTH are fetched up front -- a fetch at the
  // switch would stall the wave, and with it the whole lock-step, for a global round trip
  struct TileOps { RegRows<PrecBF16, DH> k, v; bf16x8 kb[2]; };
  auto fetch_tile = [&](int kt, TileOps& o) {  // K / V row operands, K in the dQ product's B layout
    const int k0 = kt * 32, ky = k0 + (lane & 31);
    o.k.template load_at<S16>(a.qkv, (tok0 + ky) * 3 * H + H + h * DH, ky < L);
    o.v.template load_at<S16>(a.qkv, (tok0 + ky) * 3 * H + 2 * H + h * DH, ky < L);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kr = k0 + xf_acc_row(8 * s + j, lane);
        float v = 0.f;
        if (kr < L) {
          const int64_t off = (tok0 + kr) * 3 * H + H + h * DH + (lane & 31);
          v = S16 ? (float)reinterpret_cast<const __bf16*>(a.qkv)[off] : reinterpret_cast<const float*>(a.qkv)[off];
        }
        o.kb[s][j] = (__bf16)v;
      }
  };
  auto enter_tile = [&](int kt) {
    cur_kt = kt;
    key = kt * 32 + (lane & 31);
    kin = key < L;
    kvis = kin && a.key_mask[tok0 + (kin ? key : 0)];
    all_kvis = __all(kvis);
    colmix = (uint32_t)key * kDropColMul;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
  };
  const int steps = kBwdSteps[nt - 1];
  TileOps cur, nxt;
  {
    const int code0 = kBwdSched[nt - 1][wid][0];
    int kt_a = code0 == 255 ? -1 : (code0 >> 4), kt_b = -1;
    for (int sidx = 1; sidx < steps; ++sidx) {
      const int cd = kBwdSched[nt - 1][wid][sidx];
      if (cd != 255 && (cd >> 4) != kt_a && kt_b < 0) kt_b = cd >> 4;
    }
    if (kt_a >= 0) fetch_tile(kt_a, cur);
    if (kt_b >= 0) fetch_tile(kt_b, nxt);
    if (kt_a >= 0) enter_tile(kt_a);
  }
  __syncthreads();
  for (int step = 0; step < steps; ++step) {
    const int code = kBwdSched[nt - 1][wid][step];  // (wave-uniform)
    if (code != 255) {
      const int kt = code >> 4, qb = code & 15;
      if (kt != cur_kt) {
        flush();
        cur = nxt;
        enter_tile(kt);
      }
      const int row0 = qb * 32;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      AI::tile_nreg(s, sQ, row0, cur.k.regs());
      AI::tile_nreg(dp, sDO, row0, cur.v.regs());
      const bool interior = all_kvis && qb > kt;  // every query row is after the wave's keys, every key valid
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int qi0 = row0 + 8 * g + 4 * hh;
        const float4 l4 = *reinterpret_cast<const float4*>(&sLse[qi0]);
        const float4 d4 = *reinterpret_cast<const float4*>(&sDelta[qi0]);
        const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
        uint32_t rk[4] = {0u, 0u, 0u, 0u};
        if (a.drop.on) {
          const uint4 k4 = *reinterpret_cast<const uint4*>(&sRowKey[qi0]);
          rk[0] = k4.x; rk[1] = k4.y; rk[2] = k4.z; rk[3] = k4.w;
        }
        float pr[4];
        if (interior) {
#pragma unroll
          for (int u = 0; u < 4; ++u) pr[u] = xf_exp2(fmaf(s[4 * g + u], sc, -ls[u]));
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            pr[u] = (kvis && key <= qi0 + u) ? xf_exp2(fmaf(s[4 * g + u], sc, -ls[u])) : 0.f;
        }
        float4 ds4;
        float* dsv = &ds4.x;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int r = 4 * g + u;
          const float p = pr[u];
          float keep = 1.f;
          if (a.drop.on) keep = xf_keep_scale_rc(a.drop, rk[u], colmix);
          s[r] = p * (dp[r] * keep - dl[u]);  // dS
          dp[r] = p * keep;                   // P.D
          dsv[u] = s[r];
        }
        // dS^T image: row = key (this lane), columns = the tile's queries 8g + 4hh .. + 3
        *reinterpret_cast<uint2*>(sDSw + AI::off(lane & 31, g) + 4 * hh) = xf_f32x4_to_bf16(ds4);
      }
      AI::tile_xb_tr(dv, sDO, 0, row0, dp);
      AI::tile_xb_tr(dk, sQ, 0, row0, s);
      // dQ tile [q][d] += sum_key dS^T[key][q] K[key][d]: the tile of the LDS accumulator goes through the MFMA
      // accumulator (no other wave touches query tile qb during this step: kBwdSched)
      float* dq_ptr = sDq + (row0 + 4 * hh) * DH + (lane & 31);
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = dq_ptr[((r & 3) + 8 * (r >> 2)) * DH];
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave reads back its own image
      __builtin_amdgcn_wave_barrier();
      {
        const int g16 = lane >> 4, th = g16 >> 1, li = lane & 15, qq = li >> 2, pp = li & 3;
        const int col = 16 * (g16 & 1) + 4 * pp;
        const int c = col >> 3, within = col & 7;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          union { xf_s16x4 v[2]; bf16x8 f; } av;
#pragma unroll
          for (int t2 = 0; t2 < 2; ++t2) {
            const int row = 16 * st + 8 * t2 + 4 * th + qq;
            av.v[t2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) xf_s16x4*)(sDSw + AI::off(row, c) + within));
          }
          dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av.f, cur.kb[st], dq, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dq_ptr[((r & 3) + 8 * (r >> 2)) * DH] = dq[r];
    }
    __syncthreads();  // step boundary: the next step's query tiles are dealt differently
  }
  flush();
  __syncthreads();
  for (int c = threadIdx.x; c < L * 8; c += (int)blockDim.x) {
    const int r = c >> 3, dd = (c & 7) * 4;
    float4 v = *reinterpret_cast<const float4*>(&sDq[r * DH + dd]);
    const float ks = 0.17677669529663687f;
    v.x *= ks; v.y *= ks; v.z *= ks; v.w *= ks;
    xf_st4<S16>(a.d_qkv, (tok0 + r) * 3 * H + h * DH + dd, v);
  }
}

// ---- fused backward, two ROLES per workgroup (round 4) ----------------------------------------------------------------
// The lock-step form above makes four waves share every query tile's dQ: 7-9 barriers, a read-modify-write of the fp32 dQ
// tile through LDS per step, a dS image per wave -- 2 048 workgroups of ~28 us whose waves wait half their resident cycles
// (profiles/r02_step_pmc.md). Here ONE workgroup per (batch, head) still stages Q, K, V, dO (+ delta = rowsum(dO * O), lse,
// dropout row keys, key-mask bits) once, but it has EIGHT waves in two roles that never talk to each other again:
//   waves 0-3 own KEY tiles   (S = Q K^T, query in the registers, key on the lane):  dV^T += dO^T (P.D),  dK^T += Q^T dS
//   waves 4-7 own QUERY tiles (S^T = K Q^T, key in the registers, query on the lane): dQ^T += K^T dS^T
// i.e. the loops of the dK/dV kernel and of the dQ kernel (above) side by side on the same images: every probability is
// evaluated twice (the matrix and vector pipes have the room: 5 % / 35 % busy in the lock-step form), and in exchange there
// is ONE barrier, no dQ accumulator in LDS, no dS image, no schedule table, and twice the waves per SIMD to hide latencies.
// Tiles are dealt so that the waves of a role walk the same number of tile pairs (key tile t has nt - t query tiles, query
// tile t has t + 1 key tiles): even nt: {w, nt-1-w}; odd nt: {0}, {w, nt-w}. Results: the same sums in a fixed order,
// bit-reproducible; dQ differs from the lock-step form in summation order only (its key tiles are added in ascending order).
__device__ __forceinline__ void roles_deal(int nt, int w, int nw, int out[2]) {  // key-tile indices of dK/dV wave w (-1: none)
  out[0] = out[1] = -1;
  if (nw == 8) {  // 9 ... 16 tiles (256 < L <= 512), eight waves per role: {w, nt - 1 - w} with the partner among tiles 8 ...
    out[0] = w;
    if (nt - 1 - w >= 8) out[1] = nt - 1 - w;
    return;
  }
  if (nt & 1) {
    if (w == 0) out[0] = 0;
    else if (w <= (nt - 1) / 2) { out[0] = w; out[1] = nt - w; }
  } else if (w < nt / 2) {
    out[0] = w; out[1] = nt - 1 - w;
  }
}

// NW = waves per role: 4 (up to 8 tiles: two workgroups of 8 waves per CU) or 8 (9 ... 16 tiles: the four images of a
// 512-row sequence are 128 KB of LDS -- one workgroup per CU, which then brings all 16 waves itself)
template <bool S16, int NW>
__global__ __launch_bounds__(128 * NW, 4) void attn_bwd_roles_bf16_kernel(const AttnArgs a_in) {
  AttnArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = a.H;
  int L = a.L;  // this sequence's rows (packed layout: its own length)
  const AttnBlock blk = attn_seq_block(a);
  if (!blk.valid) return;  // (whole workgroup)
  const int b = blk.by / a.A, h = blk.by % a.A;
  int64_t tok0 = (int64_t)b * a.L;
  if (a.offs) {  // packed rows: the sequence starts at its offset and is as long as it is
    const int o0 = a.offs[b];
    L = a.offs[b + 1] - o0;
    tok0 = o0;
    if (L <= 0) return;  // (whole workgroup)
  }
  const int Lp = ((L + 31) / 32) * 32, nt = Lp / 32;
  __bf16* sQ = reinterpret_cast<__bf16*>(smem_raw);
  __bf16* sDO = sQ + Lp * DH;
  __bf16* sK = sDO + Lp * DH;
  __bf16* sV = sK + Lp * DH;
  float* sLse = reinterpret_cast<float*>(sV + Lp * DH);
  float* sDelta = sLse + Lp;
  uint32_t* sRowKey = reinterpret_cast<uint32_t*>(sDelta + Lp);
  uint32_t* sBits = sRowKey + Lp;  // key mask, one bit per key (Lp / 32 + 2 words)

  const int64_t hoff = tok0 * H + h * DH;
  {
    // one round trip: the Q, K, V, dO and ctx pieces of a row chunk are loaded together (all loads of a thread's two
    // chunks before the first LDS store); delta is reduced over the lanes that hold the row's pieces
    constexpr int U = 2, PPR = S16 ? 4 : 8;  // 16-byte pieces per 32-wide row of bf16 / fp32
    const int total = Lp * PPR;
    for (int c0 = threadIdx.x; c0 < total; c0 += (int)blockDim.x * U) {
      uint4 vq[U], vk[U], vv[U], vd[U], vo[U];
      float ls[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c / PPR, pc = c % PPR;
        vq[u] = vk[u] = vv[u] = vd[u] = vo[u] = make_uint4(0u, 0u, 0u, 0u);
        ls[u] = INFINITY;
        if (c < total && r < L) {
          const int e = pc * (32 / PPR);  // first element of the piece
          const int64_t qo = (tok0 + r) * 3 * H + h * DH + e;
          vq[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.qkv, qo));
          vk[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.qkv, qo + H));
          vv[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.qkv, qo + 2 * H));
          vd[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.d_ctx, hoff + (int64_t)r * H + e));
          vo[u] = *reinterpret_cast<const uint4*>(xf_at<S16>(a.ctx, hoff + (int64_t)r * H + e));
          if (pc == 0) ls[u] = a.lse[(int64_t)blk.by * a.L + r] * kLog2e;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * (int)blockDim.x, r = c / PPR, pc = c % PPR;
        float part;
        if (S16) {
          if (c < total) {
            *reinterpret_cast<uint4*>(sQ + AI::off(r, pc)) = vq[u];
            *reinterpret_cast<uint4*>(sK + AI::off(r, pc)) = vk[u];
            *reinterpret_cast<uint4*>(sV + AI::off(r, pc)) = vv[u];
            *reinterpret_cast<uint4*>(sDO + AI::off(r, pc)) = vd[u];
          }
          const float4 d0 = xf_bf16x4_to_f32(make_uint2(vd[u].x, vd[u].y)), d1 = xf_bf16x4_to_f32(make_uint2(vd[u].z, vd[u].w));
          const float4 o0 = xf_bf16x4_to_f32(make_uint2(vo[u].x, vo[u].y)), o1 = xf_bf16x4_to_f32(make_uint2(vo[u].z, vo[u].w));
          part = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z +
                 d1.w * o1.w;
        } else {
          const float4 d4 = *reinterpret_cast<const float4*>(&vd[u]), o4 = *reinterpret_cast<const float4*>(&vo[u]);
          if (c < total) {
            const int dd = pc * 4, o = AI::off(r, dd >> 3) + (dd & 7);
            xf_store4<PrecBF16>(sQ + o, *reinterpret_cast<const float4*>(&vq[u]));
            xf_store4<PrecBF16>(sK + o, *reinterpret_cast<const float4*>(&vk[u]));
            xf_store4<PrecBF16>(sV + o, *reinterpret_cast<const float4*>(&vv[u]));
            xf_store4<PrecBF16>(sDO + o, d4);
          }
          part = d4.x * o4.x + d4.y * o4.y + d4.z * o4.z + d4.w * o4.w;
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        if (!S16) part += __shfl_xor(part, 4, 64);
        if (c < total && pc == 0) {
          sDelta[r] = part;
          sLse[r] = ls[u];
          sRowKey[r] = xf_drop_rowkey(a.drop, (uint32_t)((int64_t)blk.by * a.L + r));
        }
      }
    }
  }
  stage_key_bits(sBits, a.key_mask + tok0, Lp, L);
  __syncthreads();

  const int lane = xf_lane(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hh = lane >> 5;
  const float sc = 0.17677669529663687f * kLog2e, ks = 0.17677669529663687f;
  int mine[2];
  roles_deal(nt, wid % NW, NW, mine);
  if (wid < NW) {
    // ---- role 1: this wave's KEY tiles; walks the query tiles at or after each (attn_bwd_dkv_bf16_kernel's loop)
#pragma unroll 1
    for (int ti = 0; ti < 2; ++ti) {
      const int kt = mine[ti];
      if (kt < 0) continue;
      const int k0 = kt * 32, key = k0 + (lane & 31);
      const bool kin = key < L;
      const bool kvis = (sBits[kt] >> (lane & 31)) & 1u;
      const bool all_kvis = __builtin_amdgcn_readfirstlane(sBits[kt]) == 0xFFFFFFFFu;
      const uint32_t colmix = (uint32_t)key * kDropColMul;
      // (the key rows -- the B operands -- are re-read from the K / V images per tile: 16 registers fewer across the loop)
      f32x16 dk, dv;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
      for (int qb = kt; qb < nt; ++qb) {
        const int row0 = qb * 32;
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        AI::tile_nt(s, sQ, row0, sK, k0);
        __builtin_amdgcn_sched_barrier(0);  // (phase by phase: operand fragments of a later phase are not fetched early --
        AI::tile_nt(dp, sDO, row0, sV, k0);  //  this role has to fit the 128 registers of four waves per SIMD)
        __builtin_amdgcn_sched_barrier(0);
        const bool interior = all_kvis && qb > kt;  // every query row is after the wave's keys, every key valid
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int qi0 = row0 + 8 * g + 4 * hh;
          const float4 l4 = *reinterpret_cast<const float4*>(&sLse[qi0]);
          const float4 d4 = *reinterpret_cast<const float4*>(&sDelta[qi0]);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
          uint32_t rk[4] = {0u, 0u, 0u, 0u};
          if (a.drop.on) {
            const uint4 k4 = *reinterpret_cast<const uint4*>(&sRowKey[qi0]);
            rk[0] = k4.x; rk[1] = k4.y; rk[2] = k4.z; rk[3] = k4.w;
          }
          float pr[4];
          if (interior) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pr[u] = xf_exp2(fmaf(s[4 * g + u], sc, -ls[u]));
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              pr[u] = (kvis && key <= qi0 + u) ? xf_exp2(fmaf(s[4 * g + u], sc, -ls[u])) : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int r = 4 * g + u;
            const float p = pr[u];
            float keep = 1.f;
            if (a.drop.on) keep = xf_keep_scale_rc(a.drop, rk[u], colmix);
            s[r] = p * (dp[r] * keep - dl[u]);  // dS
            dp[r] = p * keep;                   // P.D
          }
          // keep the four groups' lse / delta / row-key reads from being hoisted together (48 registers: the loop would spill
          // at the 128 of four waves per SIMD); the other waves of the SIMD cover the LDS latency
          __builtin_amdgcn_sched_barrier(0);
        }
        AI::tile_xb_tr(dv, sDO, 0, row0, dp);
        __builtin_amdgcn_sched_barrier(0);
        AI::tile_xb_tr(dk, sQ, 0, row0, s);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kin) {  // lane = key row, 4 consecutive d per register group
        const int64_t o = (tok0 + key) * 3 * H + h * DH;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          xf_st4<S16>(a.d_qkv, o + H + 8 * g + 4 * hh,
                      make_float4(dk[4 * g] * ks, dk[4 * g + 1] * ks, dk[4 * g + 2] * ks, dk[4 * g + 3] * ks));
          xf_st4<S16>(a.d_qkv, o + 2 * H + 8 * g + 4 * hh,
                      make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]));
        }
      }
    }
  } else {
    // ---- role 2: this wave's QUERY tiles (the mirror image of the key deal: tile t has t + 1 key tiles); walks the key
    // tiles at or before each (attn_bwd_dq_bf16_kernel's loop)
#pragma unroll 1
    for (int ti = 0; ti < 2; ++ti) {
      if (mine[ti] < 0) continue;
      const int qt = nt - 1 - mine[ti];
      const int q0 = qt * 32, q = q0 + (lane & 31);
      const float lse2 = sLse[q], delta = sDelta[q];  // (rows >= L: lse = +inf -> every probability 0)
      const uint32_t rowkey = sRowKey[q];
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
      for (int kb = 0; kb <= qt; ++kb) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
        AI::tile_nt(s, sK, kb * 32, sQ, q0);
        AI::tile_nt(dp, sV, kb * 32, sDO, q0);
        const uint32_t kword = sBits[kb];
        const bool interior = kb < qt && __builtin_amdgcn_readfirstlane(kword) == 0xFFFFFFFFu;
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = xf_exp2(fmaf(s[r], sc, -lse2));
            float dpv = dp[r];
            if (a.drop.on) dpv *= xf_keep_scale_rc(a.drop, rowkey, (uint32_t)(kb * 32 + xf_acc_row(r, lane)) * kDropColMul);
            s[r] = p * (dpv - delta);
          }
        } else {
          const uint32_t kbits = kword >> (4 * hh);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kb * 32 + xf_acc_row(r, lane);
            const bool vis = key <= q && ((kbits >> ((r & 3) + 8 * (r >> 2))) & 1u);
            const float p = vis ? xf_exp2(fmaf(s[r], sc, -lse2)) : 0.f;
            float dpv = dp[r];
            if (a.drop.on) dpv *= xf_keep_scale_rc(a.drop, rowkey, (uint32_t)key * kDropColMul);
            s[r] = p * (dpv - delta);
          }
        }
        AI::tile_xb_tr(dq, sK, 0, kb * 32, s);
      }
      if (q < L) {  // dq[r] <-> (d = acc_row(r), query = lane & 31): 4 consecutive d per register group
        const int64_t o = (tok0 + q) * 3 * H + h * DH;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          xf_st4<S16>(a.d_qkv, o + 8 * g + 4 * hh,
                      make_float4(dq[4 * g] * ks, dq[4 * g + 1] * ks, dq[4 * g + 2] * ks, dq[4 * g + 3] * ks));
      }
    }
  }
}

size_t bf16_smem_roles(int L) {  // Q, dO, K, V images; lse + delta + row keys; key-mask bits
  const size_t Lp = ((size_t)L + 31) / 32 * 32;
  return 4 * Lp * DH * 2 + 3 * Lp * 4 + (Lp / 32 + 2) * sizeof(uint32_t);
}

size_t bf16_smem_fused(int L) {  // Q + dO images, fp32 dQ accumulator, lse + delta + row keys, 4 dS images
  const size_t Lp = ((size_t)L + 31) / 32 * 32;
  return 2 * Lp * DH * 2 + Lp * DH * 4 + 3 * Lp * 4 + 4 * 32 * DH * 2;
}

size_t bf16_smem_fwd(int L) {  // K + V images (aliased by the transposed-store scratch), key mask
  const int Lp = ((L + 31) / 32) * 32;
  return bf16_panel_bytes(L) + (Lp / 32 + 2) * sizeof(uint32_t);
}
size_t bf16_smem_dkv(int L) {  // Q + dO images (aliased by the scratch), lse + delta
  const int Lp = ((L + 31) / 32) * 32;
  return bf16_panel_bytes(L) + 3 * (size_t)Lp * sizeof(float);
}

template <class P, int DHT = DH>
size_t fwd_smem(int L) {
  using SM = AttnSmem<P, DHT>;
  const int Lp = ((L + 31) / 32) * 32;
  return SM::align(SM::row_img(Lp)) + SM::align(SM::t_img(Lp)) + 4 * 32 * 33 * sizeof(float) + Lp;
}
template <class P, int DHT = DH>
size_t dq_smem(int L) {
  using SM = AttnSmem<P, DHT>;
  const int Lp = ((L + 31) / 32) * 32;
  return 2 * SM::align(SM::row_img(Lp)) + SM::align(SM::t_img(Lp)) + 4 * 32 * 33 * sizeof(float) + Lp;
}
template <class P, int DHT = DH>
size_t dkv_smem(int L) {
  using SM = AttnSmem<P, DHT>;
  const int Lp = ((L + 31) / 32) * 32;
  return 2 * SM::align(SM::row_img(Lp)) + 2 * SM::align(SM::t_img(Lp)) + 4 * 32 * 33 * sizeof(float) +
         2 * (size_t)Lp * sizeof(float);
}
constexpr size_t kLdsLimit = 160 * 1024;

template <bool S16>
int launch_fwd_bf16(const AttnArgs& a, hipStream_t st) {
  static const int two_blocks = [] { const char* e = getenv("XFMR_ATTN_FWD_SPLIT"); return e ? atoi(e) : 0; }();
  if (a.offs && !(a.causal && a.L <= kSeqMaxL)) return XFMR_EUNSUPPORTED;  // packed rows: the one-workgroup forms only
  if ((!two_blocks || a.offs) && a.causal && a.L <= kSeqMaxL) {  // the one-workgroup forms walk the causal triangle only
    const size_t sf = bf16_smem_fwd_seq(a.L);
    if (hipFuncSetAttribute((const void*)attn_fwd_seq_bf16_kernel<S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sf) != hipSuccess)
      return XFMR_EHIP;
    hipLaunchKernelGGL((attn_fwd_seq_bf16_kernel<S16>), dim3((unsigned)(a.A * ((a.B + 7) / 8) * 8)), dim3(256), sf, st,
                       a);
    XF_LAUNCH_CHECK();
    return XFMR_OK;
  }
  dim3 grid((unsigned)(((a.L + 127) / 128) * a.A * ((a.B + 7) / 8) * 8));  // see attn_block
  const size_t sm = bf16_smem_fwd(a.L);
  if (sm > kLdsLimit) return XFMR_EUNSUPPORTED;
  if (hipFuncSetAttribute((const void*)attn_fwd_bf16_kernel<S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sm) != hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<S16>), grid, dim3(256), sm, st, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
template <bool S16>
int launch_bwd_bf16(const AttnArgs& a, hipStream_t st) {
  static const int two_kernels = [] { const char* e = getenv("XFMR_ATTN_BWD_SPLIT"); return e ? atoi(e) : 0; }();
  // the one-workgroup forms: round 1's four-wave lock-step form, or -- XFMR_ATTN_BWD_FORM=roles, read per call -- round 4's
  // "roles" (eight waves, two roles, one barrier). MEASURED, alternating runs on one box, batch 512: 113.9 / 112.1 us per
  // layer for roles against 111.1 / 110.7 for lock-step (3.330 / 3.333 vs 3.315 / 3.302 ms per step): the barriers and
  // the dQ read-modify-write it removes are paid back by evaluating every probability twice (the vector pipe: 28 tile
  // pairs x ~2 800 issue cycles per (batch, head) against 28 x ~2 000) -- not the default. DESIGN.md section 4.
  const char* form = getenv("XFMR_ATTN_BWD_FORM");
  const bool roles = form && form[0] == 'r';
  if (a.offs && !(a.causal && a.L <= kSeqMaxL)) return XFMR_EUNSUPPORTED;  // packed rows: the one-workgroup forms only
  // 256 < L <= 512 (BASELINE config 5): the two-role form is the one-workgroup backward there (the lock-step form's schedule
  // table stops at eight tiles): Q, K, V, dO of 512 rows are 128 KB of LDS -- one workgroup of eight waves per CU
  if ((!two_kernels || a.offs) && (roles || a.L > kFusedMaxL) && a.causal && a.L <= kSeqMaxL) {
    const size_t sr = bf16_smem_roles(a.L);
    const dim3 grid((unsigned)(a.A * ((a.B + 7) / 8) * 8));
    if (a.L > kFusedMaxL) {
      if (hipFuncSetAttribute((const void*)attn_bwd_roles_bf16_kernel<S16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sr) != hipSuccess)
        return XFMR_EHIP;
      hipLaunchKernelGGL((attn_bwd_roles_bf16_kernel<S16, 8>), grid, dim3(1024), sr, st, a);
    } else {
      if (hipFuncSetAttribute((const void*)attn_bwd_roles_bf16_kernel<S16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sr) != hipSuccess)
        return XFMR_EHIP;
      hipLaunchKernelGGL((attn_bwd_roles_bf16_kernel<S16, 4>), grid, dim3(512), sr, st, a);
    }
    XF_LAUNCH_CHECK();
    return XFMR_OK;
  }
  const size_t sf = bf16_smem_fused(a.L);
  if ((!two_kernels || a.offs) && a.causal && a.L <= kFusedMaxL && sf <= kLdsLimit) {
    if (hipFuncSetAttribute((const void*)attn_bwd_fused_bf16_kernel<S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sf) != hipSuccess)
      return XFMR_EHIP;
    hipLaunchKernelGGL((attn_bwd_fused_bf16_kernel<S16>), dim3((unsigned)(a.A * ((a.B + 7) / 8) * 8)), dim3(256), sf,
                       st, a);
    XF_LAUNCH_CHECK();
    return XFMR_OK;
  }
  dim3 grid((unsigned)(((a.L + 127) / 128) * a.A * ((a.B + 7) / 8) * 8));  // see attn_block
  const size_t s1 = bf16_smem_fwd(a.L), s2 = bf16_smem_dkv(a.L);
  if (s1 > kLdsLimit || s2 > kLdsLimit) return XFMR_EUNSUPPORTED;
  if (hipFuncSetAttribute((const void*)attn_bwd_dq_bf16_kernel<S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)s1) != hipSuccess ||
      hipFuncSetAttribute((const void*)attn_bwd_dkv_bf16_kernel<S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)s2) != hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<S16>), grid, dim3(256), s1, st, a);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<S16>), grid, dim3(256), s2, st, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

// The generic kernels: fp32 policy at head size 32, and both policies at head size 64 (operands fp32, or bf16 when S16).
template <class P, int DHT, bool S16>
int launch_fwd_generic(const AttnArgs& a, hipStream_t st) {
  dim3 grid((a.L + 127) / 128, a.B * a.A);
  const size_t sm = fwd_smem<P, DHT>(a.L);
  if (sm > kLdsLimit) return XFMR_EUNSUPPORTED;
  if (hipFuncSetAttribute((const void*)attn_fwd_kernel<P, DHT, S16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) !=
      hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL((attn_fwd_kernel<P, DHT, S16>), grid, dim3(256), sm, st, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
template <class P, int DHT, bool S16>
int launch_bwd_generic(const AttnArgs& a, hipStream_t st) {
  dim3 grid((a.L + 127) / 128, a.B * a.A);
  const size_t s1 = dq_smem<P, DHT>(a.L), s2 = dkv_smem<P, DHT>(a.L);
  if (s1 > kLdsLimit || s2 > kLdsLimit) return XFMR_EUNSUPPORTED;
  if (hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<P, DHT, S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)s1) != hipSuccess ||
      hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<P, DHT, S16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)s2) != hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<P, DHT, S16>), grid, dim3(256), s1, st, a);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<P, DHT, S16>), grid, dim3(256), s2, st, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
// precision x storage x head size -> kernel family
int dispatch_fwd(const AttnArgs& a, int precision, bool s16, hipStream_t st) {
  const int dh = a.H / a.A;
  if (a.offs && !(precision == XFMR_PREC_BF16 && dh == 32)) return XFMR_EUNSUPPORTED;  // packed rows: production kernels only
  if (precision == XFMR_PREC_BF16) {
    if (dh == 32) return s16 ? launch_fwd_bf16<true>(a, st) : launch_fwd_bf16<false>(a, st);
    return s16 ? launch_fwd_generic<PrecBF16, 64, true>(a, st) : launch_fwd_generic<PrecBF16, 64, false>(a, st);
  }
  if (precision == XFMR_PREC_F32 && !s16)
    return dh == 32 ? launch_fwd_generic<PrecF32, 32, false>(a, st) : launch_fwd_generic<PrecF32, 64, false>(a, st);
  return XFMR_EINVAL;
}
int dispatch_bwd(const AttnArgs& a, int precision, bool s16, hipStream_t st) {
  const int dh = a.H / a.A;
  if (a.offs && !(precision == XFMR_PREC_BF16 && dh == 32)) return XFMR_EUNSUPPORTED;
  if (precision == XFMR_PREC_BF16) {
    if (dh == 32) return s16 ? launch_bwd_bf16<true>(a, st) : launch_bwd_bf16<false>(a, st);
    return s16 ? launch_bwd_generic<PrecBF16, 64, true>(a, st) : launch_bwd_generic<PrecBF16, 64, false>(a, st);
  }
  if (precision == XFMR_PREC_F32 && !s16)
    return dh == 32 ? launch_bwd_generic<PrecF32, 32, false>(a, st) : launch_bwd_generic<PrecF32, 64, false>(a, st);
  return XFMR_EINVAL;
}

int check_shape(int B, int L, int A, int H) {
  if (B <= 0 || L <= 0 || A <= 0 || H <= 0) return XFMR_EINVAL;
  if (H != A * 32 && H != A * 64) return XFMR_EUNSUPPORTED;  // head size 32 (the production kernels) or 64 (generic kernels)
  return XFMR_OK;
}

}  // namespace

extern "C" {

int xf_attn_fwd_ex(const void* qkv, const uint8_t* key_mask, void* ctx, float* lse, int32_t B, int32_t L, int32_t A,
                   int32_t H, float dropout_p, XfSeed seed, uint32_t site, int32_t precision, bool s16,
                   bool causal, hipStream_t st, const int32_t* seq_offsets) {
  if (!qkv || !key_mask || !ctx || !lse) return XFMR_EINVAL;
  if (int rc = check_shape(B, L, A, H)) return rc;
  if (!xf_aligned16(qkv) || !xf_aligned16(ctx)) return XFMR_EALIGN;
  AttnArgs a{};
  a.qkv = (const float*)qkv; a.key_mask = key_mask; a.ctx = (float*)ctx; a.lse = lse; a.B = B; a.L = L; a.A = A;
  a.H = H; a.causal = causal; a.offs = seq_offsets;
  a.drop = xf_make_dropout(dropout_p, seed, site);
  return dispatch_fwd(a, precision, s16, st);
}

int xfmr_attn_fwd(const float* qkv, const uint8_t* key_mask, float* ctx, float* lse, int32_t B, int32_t L,
                  int32_t A, int32_t H, float dropout_p, uint64_t seed, uint32_t site, int32_t precision,
                  void* stream) {
  return xf_attn_fwd_ex(qkv, key_mask, ctx, lse, B, L, A, H, dropout_p, seed, site, precision, false, true,
                        (hipStream_t)stream);
}
int xfmr_attn_fwd_mode(const float* qkv, const uint8_t* key_mask, float* ctx, float* lse, int32_t B, int32_t L,
                       int32_t A, int32_t H, float dropout_p, uint64_t seed, uint32_t site, int32_t precision,
                       int32_t attn_mode, void* stream) {
  if (attn_mode != XFMR_ATTN_CAUSAL && attn_mode != XFMR_ATTN_BIDIRECTIONAL) return XFMR_EINVAL;
  return xf_attn_fwd_ex(qkv, key_mask, ctx, lse, B, L, A, H, dropout_p, seed, site, precision, false,
                        attn_mode == XFMR_ATTN_CAUSAL, (hipStream_t)stream);
}

int xf_attn_bwd_ex(const void* qkv, const uint8_t* key_mask, const void* ctx, const float* lse, const void* d_ctx,
                   void* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p, XfSeed seed,
                   uint32_t site, int32_t precision, bool s16, bool causal, hipStream_t st, const int32_t* seq_offsets) {
  if (!qkv || !key_mask || !ctx || !lse || !d_ctx || !d_qkv) return XFMR_EINVAL;
  if (int rc = check_shape(B, L, A, H)) return rc;
  if (!xf_aligned16(qkv) || !xf_aligned16(ctx) || !xf_aligned16(d_ctx) || !xf_aligned16(d_qkv)) return XFMR_EALIGN;
  AttnArgs a{};
  a.qkv = (const float*)qkv; a.key_mask = key_mask; a.ctx = (float*)const_cast<void*>(ctx);
  a.lse = const_cast<float*>(lse); a.d_ctx = (const float*)d_ctx; a.d_qkv = (float*)d_qkv;
  a.B = B; a.L = L; a.A = A; a.H = H; a.causal = causal; a.offs = seq_offsets;
  a.drop = xf_make_dropout(dropout_p, seed, site);
  return dispatch_bwd(a, precision, s16, st);
}

int xfmr_attn_bwd(const float* qkv, const uint8_t* key_mask, const float* ctx, const float* lse,
                  const float* d_ctx, float* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p,
                  uint64_t seed, uint32_t site, int32_t precision, void* stream) {
  return xf_attn_bwd_ex(qkv, key_mask, ctx, lse, d_ctx, d_qkv, B, L, A, H, dropout_p, seed, site, precision, false,
                        true, (hipStream_t)stream);
}
int xfmr_attn_bwd_mode(const float* qkv, const uint8_t* key_mask, const float* ctx, const float* lse,
                       const float* d_ctx, float* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p,
                       uint64_t seed, uint32_t site, int32_t precision, int32_t attn_mode, void* stream) {
  if (attn_mode != XFMR_ATTN_CAUSAL && attn_mode != XFMR_ATTN_BIDIRECTIONAL) return XFMR_EINVAL;
  return xf_attn_bwd_ex(qkv, key_mask, ctx, lse, d_ctx, d_qkv, B, L, A, H, dropout_p, seed, site, precision, false,
                        attn_mode == XFMR_ATTN_CAUSAL, (hipStream_t)stream);
}

}  // extern "C"
