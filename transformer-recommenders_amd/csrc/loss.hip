// Fused in-batch sampled loss: all seven heads + LogitsStatistics + dL/dquery in one pass over the
// (never materialised) logits  [ rowdot(q, E[pos]) | Q E[neg]^T ].
//
// Structure (flash-attention shaped, with "V = K = E_neg"):
//   * a workgroup = 4 waves owns 128 queries; each wave keeps its 32 query rows in registers as the B
//     operand and walks tiles of 64 negatives whose table rows are GATHERED by item id straight into
//     LDS (fp32 HBM/L2 rows -> registers -> MFMA element type), next tile's gather in flight during the
//     current tile's math;
//   * S^T = E_neg Q^T is produced with the negative in the accumulator registers and the query on the
//     lane, so every per-query reduction (false-negative mask counts, online log-sum-exp, softplus /
//     hinge / relu sums, statistics) is lane-local;
//   * the per-element loss-gradient weights w (softmax numerators / sigmoids / indicators) go back into
//     the matrix core as the B operand straight from the accumulator registers:
//     dQ^T += E_neg^T w^T  (second image of the tile, [h][j], conflict-free for the b64 fragment reads);
//   * the negatives axis is split across workgroups (grid.y) to fill 256 CUs; a combine kernel merges the
//     split partials per query, finishes the seven row losses and the gradient row, and a final
//     single-workgroup kernel reduces them deterministically.
#include "internal.h"
#include "loss_common.h"

namespace {

// ---- compaction of valid positions (replaces the boolean-mask indexing of models.py:390-404) ------
// Two fully parallel kernels, order-preserving (the reference's boolean indexing keeps row-major order):
//   count: workgroup b counts the valid positions / queries of its 1024 positions -> blockcnt[b]
//   write: workgroup b sums blockcnt[0..b) (at most a few hundred ints), scans its own 1024 flags with
//          wave ballots, and writes the compacted item lists; the last workgroup publishes the totals.
constexpr int PREP = 1024;
__global__ __launch_bounds__(PREP) void prepare_count_kernel(const uint8_t* key_mask, const int64_t* pos_idx, int T,
                                                             int2* blockcnt) {
  __shared__ int sv[PREP / 64], sq[PREP / 64];
  const int i = blockIdx.x * PREP + threadIdx.x;
  const bool v = i < T && key_mask[i] != 0;
  const bool q = v && pos_idx[i] != 0;
  const int nv = __popcll(__ballot(v)), nq = __popcll(__ballot(q));
  if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = nv; sq[threadIdx.x >> 6] = nq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int w = 0; w < PREP / 64; ++w) { a += sv[w]; b += sq[w]; }
    blockcnt[blockIdx.x] = make_int2(a, b);
  }
}
__global__ __launch_bounds__(PREP) void prepare_write_kernel(const uint8_t* key_mask, const int64_t* pos_idx,
                                                             const int64_t* neg_idx, int T, int64_t n_rows,
                                                             const int2* blockcnt, int* counts, int* hist, int* qrow,
                                                             int* qpos) {
  __shared__ int sv[PREP / 64], sq[PREP / 64], base[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int i = blockIdx.x * PREP + tid;
  const bool v = i < T && key_mask[i] != 0;
  const int64_t pi = v ? pos_idx[i] : 0;
  const bool q = v && pi != 0;
  const unsigned long long bv = __ballot(v), bq = __ballot(q);
  if (lane == 0) { sv[w] = __popcll(bv); sq[w] = __popcll(bq); }
  if (tid == 0) {
    int a = 0, b = 0;
    for (int k = 0; k < (int)blockIdx.x; ++k) { a += blockcnt[k].x; b += blockcnt[k].y; }
    base[0] = a; base[1] = b;
  }
  __syncthreads();
  int ov = base[0], oq = base[1];
  for (int k = 0; k < w; ++k) { ov += sv[k]; oq += sq[k]; }
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  ov += __popcll(bv & below);
  oq += __popcll(bq & below);
  if (v) {
    if (neg_idx) {  // multiplicity of the position's negative item (integer atomics: deterministic)
      int64_t ni = neg_idx[i];
      if (ni < 0 || ni >= n_rows) ni = 0;
      atomicAdd(&hist[ni], 1);
    }
    if (q) {
      qrow[oq] = i;
      qpos[oq] = (int)((pi < 0 || pi >= n_rows) ? 0 : pi);
    }
  }
  if (blockIdx.x == gridDim.x - 1 && tid == PREP - 1) {
    counts[0] = ov + (v ? 1 : 0);
    counts[1] = oq + (q ? 1 : 0);
  }
}

// ---- list form: queries already compacted (EmbedLoss.forward(query_embed, candidates)) ------------
__global__ void prepare_lists_kernel(const int64_t* pos_items, const int64_t* neg_items, int Np, int N,
                                     int64_t n_rows, int* counts, int* hist, int* qrow, int* qpos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    counts[0] = N;
    counts[1] = Np;
  }
  if (i < Np) {
    int64_t p = pos_items[i];
    if (p < 0 || p >= n_rows) p = 0;
    qrow[i] = i;
    qpos[i] = (int)p;
  }
  if (neg_items && i < N) {
    int64_t n = neg_items[i];
    if (n < 0 || n >= n_rows) n = 0;
    atomicAdd(&hist[n], 1);
  }
}

// ---- distinct negative items (ascending id) with their multiplicities, from the histogram over the catalogue --
// Same two-kernel, order-preserving compaction as above, over items instead of positions.
__global__ __launch_bounds__(PREP) void distinct_count_kernel(const int* hist, int64_t n_rows, int* blockcnt) {
  __shared__ int sv[PREP / 64];
  const int64_t i = (int64_t)blockIdx.x * PREP + threadIdx.x;
  const bool v = i < n_rows && hist[i] > 0;
  const int nv = __popcll(__ballot(v));
  if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = nv;
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0;
    for (int w = 0; w < PREP / 64; ++w) a += sv[w];
    blockcnt[blockIdx.x] = a;
  }
}
__global__ __launch_bounds__(PREP) void distinct_write_kernel(const int* hist, int64_t n_rows, const int* blockcnt,
                                                              const float* rnorm, int* counts, int* neg_item,
                                                              float* neg_rc, float* neg_mult) {
  __shared__ int sv[PREP / 64], base;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t i = (int64_t)blockIdx.x * PREP + tid;
  const int h = i < n_rows ? hist[i] : 0;
  const bool v = h > 0;
  const unsigned long long bv = __ballot(v);
  if (lane == 0) sv[w] = __popcll(bv);
  if (tid == 0) {
    int a = 0;
    for (int k = 0; k < (int)blockIdx.x; ++k) a += blockcnt[k];
    base = a;
  }
  __syncthreads();
  int ov = base;
  for (int k = 0; k < w; ++k) ov += sv[k];
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  ov += __popcll(bv & below);
  if (v) {
    neg_item[ov] = (int)i;
    neg_rc[ov] = rnorm[i];
    neg_mult[ov] = (float)h;
  }
  if (blockIdx.x == gridDim.x - 1 && tid == PREP - 1) counts[2] = ov + (v ? 1 : 0);
}

// ---- main kernel -------------------------------------------------------------------------------------
// H > 256 (the reference's default d_model 384): the dQ product runs as H / 128 column parts (grid.z) -- the tile image
// [64][H] for S^T is whole, the transposed image and the gradient accumulators cover 128 columns: 134 KB of LDS and 64
// accumulator registers in the fp32 policy instead of 203 KB / 192. Scores and weights are recomputed per part.
// H is the kernel's TEMPLATE width: the rows' real width Hr = a.H (their stride too) may be any multiple of 32 up to it
// -- columns Hr .. H - 1 are zeros in every image and register operand, which changes no dot product, norm or gradient.
// That is how every d_model the reference can be configured with (models.py:22-48: any hidden_size) reaches this kernel:
// the width is rounded up to the next instantiation (64, 128, 256, 384, 512, 768, 1024; the fp32 policy up to 512: its
// tile image is 4 bytes per element, and at 512 the dQ parts are 64 columns wide so that both images fit 160 KB).
template <class P, int H>
constexpr int loss_main_hparts() { return H <= 256 ? 1 : (sizeof(typename P::elem) == 4 && H >= 512) ? H / 64 : H / 128; }
template <class P, int H, bool ALL>
__global__ __launch_bounds__(256, (H <= 128 ? 2 : 1)) void loss_main_kernel(LossArgs a) {
  using elem = typename P::elem;
  constexpr int LDE = xf_ld<P>(H);
  constexpr int NPASS = 2 * (H / 64);  // staging passes: 32 rows x 64 columns per pass per workgroup
  constexpr int HP = loss_main_hparts<P, H>(), HW = H / HP, NO = HW / 32;
  const int Hr = a.H;
  const int hpart = HP > 1 ? (int)blockIdx.z : 0;
  // With every head evaluated the epilogue needs the registers: the wave's query rows then live in LDS
  // (B operand read like the A operand) so the kernel still fits 2 waves/SIMD (<= 256 registers).
  constexpr bool Q_IN_LDS = ALL && (P::kId == XFMR_PREC_BF16) && (H <= 128);
  constexpr size_t E_BYTES = BN * LDE * sizeof(elem), ET_BYTES = HW * LDT * sizeof(elem);
  constexpr size_t SCR_BYTES = 4 * 32 * 33 * sizeof(float);
  constexpr size_t MAIN_BYTES = (E_BYTES + ET_BYTES > SCR_BYTES) ? E_BYTES + ET_BYTES : SCR_BYTES;
  constexpr size_t Q_BYTES = Q_IN_LDS ? QB * LDE * sizeof(elem) : 0;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_BYTES + Q_BYTES];
  elem* const sE = reinterpret_cast<elem*>(smem);
  elem* const sET = reinterpret_cast<elem*>(smem + E_BYTES);
  float* const sScratch = reinterpret_cast<float*>(smem);  // epilogue only: aliases the tile images
  elem* const sQ = reinterpret_cast<elem*>(smem + MAIN_BYTES);
  __shared__ __attribute__((aligned(16))) int sNid[BN];
  __shared__ __attribute__((aligned(16))) float sRc[BN];
  __shared__ __attribute__((aligned(16))) float sMul[BN];  // multiplicity of each column's item (LossArgs)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, hh = lane >> 5;
  const int Nq = a.counts[1];
  const int N = (a.mode == XFMR_NEG_CATALOG) ? (int)a.n_rows : a.counts[2];  // columns = distinct negative items
  const int qb0 = blockIdx.x * QB;
  if (qb0 >= Nq) return;
  const int ntiles = (N + BN - 1) / BN;
  const int split = blockIdx.y;
  const int t_beg = (int)((int64_t)ntiles * split / a.nsplit);
  const int t_end = (int)((int64_t)ntiles * (split + 1) / a.nsplit);

  // ---- per-query prologue -----------------------------------------------------------------------
  const int qi = qb0 + wid * 32 + (lane & 31);
  const bool qvalid = qi < Nq;
  const int row = qvalid ? a.qrow[qi] : 0;
  const int pos_item = qvalid ? a.qpos[qi] : -2;
  RegRows<P, H> qreg;
  qreg.load_n(a.tok + (int64_t)row * Hr, qvalid, Hr);
  float pos_dot, qq = 0.f;
  {
    if (H > 384) {  // (a second H-wide register row beside Q would not fit: the positive's dot product in 128-column pieces)
      pos_dot = 0.f;
      for (int c0 = 0; c0 < Hr; c0 += 128) {
        RegRows<P, 128> qa, pa;
        qa.load_n(a.tok + (int64_t)row * Hr + c0, qvalid, Hr - c0);
        pa.load_n(a.table + (int64_t)(qvalid ? pos_item : 0) * Hr + c0, qvalid, Hr - c0);
        pos_dot += qa.dot_partial(pa);
      }
    } else {
      RegRows<P, H> preg;
      preg.load_n(a.table + (int64_t)(qvalid ? pos_item : 0) * Hr, qvalid, Hr);
      pos_dot = qreg.dot_partial(preg);
    }
    pos_dot += xf_half_swap(pos_dot);
    if (qvalid) {
      const float* pr = a.tok + (int64_t)row * Hr + hh * (Hr / 2);
#pragma unroll 4
      for (int c = 0; c < Hr / 2; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(pr + c);
        qq += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
    }
    qq += xf_half_swap(qq);
  }
  if (Q_IN_LDS) qreg.store_image(sQ + (wid * 32 + (lane & 31)) * LDE);  // visible after the first barrier
  const float rq = 1.f / fmaxf(sqrtf(qq), 1e-8f);
  const float rcpos = qvalid ? a.rnorm[pos_item] : 1.f;
  const float cpos = pos_dot * rq * rcpos;
  const float chinge = pos_dot * (1.f - a.margin);
  const float sc2 = a.scale * kLog2e;
  const float z2pos = pos_dot * sc2;
  const bool mask_fn = a.mask_fn != 0, catalog = (a.mode == XFMR_NEG_CATALOG);
  const int head = a.train_head;
  const bool cos_head = head <= XFMR_LOSS_CONTRASTIVE;
  const bool do_grad = a.need_grad && head != XFMR_LOSS_ALIGNMENT;

  RowState st{0.f, z2pos, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, INFINITY, -INFINITY, 0.f};
  RowConst kc{pos_dot, cpos, chinge, sc2, rq, a.margin, pos_item, head, mask_fn, catalog, cos_head};
  kc.hard = a.tau != nullptr;
  if (kc.hard) {
    const float4 th = a.tau[qvalid ? qi : 0];
    kc.tau_d = th.x; kc.rho_d = th.y; kc.tau_c = th.z; kc.rho_c = th.w;
  }
  const bool dump = a.dump != nullptr;
  if (dump && split == 0 && lane < 32 && qvalid) a.qinfo[qi] = make_float2(pos_dot, rq);
  f32x16 o[NO];
#pragma unroll
  for (int i = 0; i < NO; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

  // ---- tile staging -----------------------------------------------------------------------------
  // lane -> (row pair rho = lane&3, 16-byte chunk c = lane>>2); wave w -> rows 8w..8w+7 of a 32-row block;
  // pass p -> row block p&1, column block p>>1. ET writes then land 2-way conflicted at worst.
  float4 pre[NPASS][2];
  int pre_nid = -1;
  float pre_rc = 0.f, pre_mu = 1.f;
  const int st_rho = lane & 3, st_c = lane >> 2;
  auto item_of = [&](int j) -> int {
    if (j >= N) return -1;
    return catalog ? j : a.neg_item[j];
  };
  auto prefetch = [&](int tile) {
    const int j0 = tile * BN;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int cc = (p >> 1) * 64 + st_c * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = (p & 1) * 32 + wid * 8 + 2 * st_rho + u;
        const int it = item_of(j0 + r);
        pre[p][u] = (it >= 0 && cc < Hr) ? *reinterpret_cast<const float4*>(a.table + (int64_t)it * Hr + cc)
                                        : make_float4(0, 0, 0, 0);
      }
    }
    if (tid < BN) {
      pre_nid = item_of(j0 + tid);
      pre_rc = pre_nid >= 0 ? a.rnorm[pre_nid] : 0.f;
      pre_mu = (pre_nid >= 0 && a.neg_mult && !catalog) ? a.neg_mult[j0 + tid] : 1.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int cc = (p >> 1) * 64 + st_c * 4;
      const int r0 = (p & 1) * 32 + wid * 8 + 2 * st_rho;
      xf_store4<P>(sE + r0 * LDE + cc, pre[p][0]);
      xf_store4<P>(sE + (r0 + 1) * LDE + cc, pre[p][1]);
      const int ct = cc - hpart * HW;  // column inside this part's transposed image (HP == 1: all of them)
      if (HP > 1 && (ct < 0 || ct >= HW)) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) xf_store2<P>(sET + (ct + j) * LDT + r0, xf_get(pre[p][0], j), xf_get(pre[p][1], j));
    }
    if (tid < BN) {
      sNid[tid] = pre_nid;
      sRc[tid] = pre_rc;
      sMul[tid] = pre_mu;
    }
  };

  // PREFETCH_ACROSS: keep the next tile's gather in registers while this tile is multiplied (1 workgroup/CU
  // variants); otherwise gather-then-commit back to back and let the second resident workgroup of the CU
  // cover the latency (2 waves/SIMD variants: the 32 staging registers are not live across the math).
  constexpr bool PREFETCH_ACROSS = (H > 128 && H <= 256);  // (H = 384: the 96 staging registers are not kept live)
  // H > 384: the staging registers of a whole tile (8 per 64 columns: 128 at H = 1024, beside 256 of Q) are too many --
  // gather and commit four passes at a time
  constexpr bool STAGE_DIRECT = H > 384;
  auto stage_direct = [&](int tile) {
    const int j0 = tile * BN;
#pragma unroll 1
    for (int p0 = 0; p0 < NPASS; p0 += 4) {
      float4 t[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = p0 + q, cc = (p >> 1) * 64 + st_c * 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int it = item_of(j0 + (p & 1) * 32 + wid * 8 + 2 * st_rho + u);
          t[q][u] = (it >= 0 && cc < Hr) ? *reinterpret_cast<const float4*>(a.table + (int64_t)it * Hr + cc)
                                         : make_float4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = p0 + q, cc = (p >> 1) * 64 + st_c * 4;
        const int r0 = (p & 1) * 32 + wid * 8 + 2 * st_rho;
        xf_store4<P>(sE + r0 * LDE + cc, t[q][0]);
        xf_store4<P>(sE + (r0 + 1) * LDE + cc, t[q][1]);
        const int ct = cc - hpart * HW;
        if (HP > 1 && (ct < 0 || ct >= HW)) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) xf_store2<P>(sET + (ct + j) * LDT + r0, xf_get(t[q][0], j), xf_get(t[q][1], j));
      }
    }
    if (tid < BN) {
      const int nid = item_of(j0 + tid);
      sNid[tid] = nid;
      sRc[tid] = nid >= 0 ? a.rnorm[nid] : 0.f;
      sMul[tid] = (nid >= 0 && a.neg_mult && !catalog) ? a.neg_mult[j0 + tid] : 1.f;
    }
  };
  if (PREFETCH_ACROSS && t_beg < t_end) prefetch(t_beg);
  for (int tile = t_beg; tile < t_end; ++tile) {
    __syncthreads();
    if (STAGE_DIRECT) stage_direct(tile);
    else {
      if (!PREFETCH_ACROSS) prefetch(tile);
      commit();
    }
    __syncthreads();
    if (PREFETCH_ACROSS && tile + 1 < t_end) prefetch(tile + 1);

#pragma unroll 1
    for (int sb = 0; sb < BN / 32; ++sb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      if (Q_IN_LDS) P::tile_nt(s, sE, LDE, sb * 32, sQ, LDE, wid * 32, H);
      else P::tile_nreg(s, sE, LDE, sb * 32, qreg.regs(), H);

      if (dump) {  // logits of this sub-block, as the epilogue would see them (tie resolution happens in the reader)
        if (qvalid && hpart == 0) {
          float* drow = a.dump + (int64_t)qi * a.dump_ld + tile * BN + sb * 32 + 4 * hh;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (tile * BN + sb * 32 + 8 * g + 4 * hh < a.dump_ld)
              *reinterpret_cast<float4*>(drow + 8 * g) = make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]);
        }
        continue;
      }
      loss_epilogue<ALL, true>(s, st, o, kc, &sNid[sb * 32], &sRc[sb * 32], &sMul[sb * 32], hh);
      if (do_grad) {
#pragma unroll
        for (int i = 0; i < NO; ++i) P::tile_xb(o[i], sET, LDT, i * 32, sb * 32, s);
      }
    }
  }

  if (dump) return;
  // ---- write the (split, query) partial -----------------------------------------------------------
  write_partial(merge_halves(st), a.part + ((int64_t)split * a.T + qi) * REC, lane < 32 && qvalid && hpart == 0, pos_dot,
                rq, qq);
  if (do_grad) {
    __syncthreads();  // sScratch aliases the tile images other waves may still be reading
    float* base = a.partO + (int64_t)split * a.T * Hr;
#pragma unroll
    for (int i = 0; i < NO; ++i)
      if ((hpart * NO + i) * 32 < Hr)  // (column blocks past the real width hold zeros)
        xf_store_tile_T(sScratch + wid * 32 * 33, o[i], 1.f, base + (hpart * NO + i) * 32, Hr, qb0 + wid * 32, Nq);
  }
}

// bf16 copy of the frozen table (gather source of the LDS-DMA path) + per-item inverse norms
__global__ void table_prepare_kernel(const float* table, float* rnorm, __bf16* tbf, int64_t rows, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float e = table[row * H + c];
    s += e * e;
    if (tbf) tbf[row * H + c] = (__bf16)e;
  }
  s = xf_wave_sum(s);
  if (lane == 0 && rnorm) rnorm[row] = 1.f / fmaxf(sqrtf(s), 1e-8f);
}

// ---- num_hard_negatives: per-row top-k thresholds from the dumped logits ------------------------------------
// One workgroup per query. The k-th largest counted logit is found by a 4-pass, 8-bits-per-pass radix select over
// order-preserving keys (LDS histogram); elements equal to the threshold share the weight rho = (k - #greater) /
// #equal -- the loss depends on the multiset of selected logits only, so this equals any tie-breaking of torch.topk
// (in-batch negatives repeat items, i.e. ties are the rule, not the exception).
__device__ __forceinline__ unsigned hard_key(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float hard_unkey(unsigned k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}
template <class F>
__device__ void radix_select(F get, int N, int k, unsigned* hist, int* sh, float& tau, float& rho) {
  // get(j, &value) -> weight of column j among the counted negatives (its item's multiplicity, 0 = not counted)
  const int tid = threadIdx.x;
  if (tid == 0) sh[3] = 0;
  __syncthreads();
  int n = 0;
  for (int j = tid; j < N; j += 256) {
    float v;
    n += get(j, v);
  }
  if (n) atomicAdd(&sh[3], n);
  __syncthreads();
  if (sh[3] <= k) {  // fewer counted negatives than k: all of them stay (losses.py:318-329)
    tau = -INFINITY; rho = 1.f;
    return;
  }
  unsigned prefix = 0;
  int need = k;
  for (int p = 3; p >= 0; --p) {
    hist[tid] = 0;
    __syncthreads();
    const int shift = 8 * p;
    for (int j = tid; j < N; j += 256) {
      float v;
      const int wgt = get(j, v);
      if (!wgt) continue;
      const unsigned key = hard_key(v);
      if (p == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8)))
        atomicAdd(&hist[(key >> shift) & 255u], (unsigned)wgt);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, d = 255;
      for (; d > 0; --d) {
        if (cum + (int)hist[d] >= need) break;
        cum += (int)hist[d];
      }
      sh[0] = d; sh[1] = need - cum; sh[2] = (int)hist[d];
    }
    __syncthreads();
    prefix |= (unsigned)sh[0] << shift;
    need = sh[1];
    __syncthreads();
  }
  tau = hard_unkey(prefix);
  rho = (float)need / (float)sh[2];
}

struct SelectArgs {
  const float* dump; int64_t dump_ld; const float2* qinfo; const int* counts; const int* neg_item;
  const float* neg_rc; const float* neg_mult; const float* rnorm; const int* qpos; float4* tau;
  int mode, mask_fn, k; int64_t n_rows;
};
__global__ __launch_bounds__(256) void hard_select_kernel(SelectArgs a) {
  __shared__ unsigned hist[256];
  __shared__ int sh[4];
  const int qi = blockIdx.x;
  if (qi >= a.counts[1]) return;
  const bool catalog = a.mode == XFMR_NEG_CATALOG;
  const int N = catalog ? (int)a.n_rows : a.counts[2];       // columns (distinct items)
  const int Ncols = catalog ? (int)a.n_rows : a.counts[0];   // negative columns of the reference's logits
  float4 out = make_float4(-INFINITY, 1.f, -INFINITY, 1.f);
  if (a.k > 0 && a.k < Ncols + 1) {  // losses.py:312-316: k >= number of columns (1 + N) switches the restriction off
    const int pos_item = a.qpos[qi];
    const float2 qf = a.qinfo[qi];
    const float pos_dot = qf.x, rq = qf.y;
    const float cpos = pos_dot * rq * a.rnorm[pos_item];
    const float* row = a.dump + (int64_t)qi * a.dump_ld;
    const bool mask_fn = a.mask_fn != 0;
    auto get_d = [&](int j, float& v) {
      const int it = catalog ? j : a.neg_item[j];
      const bool same = it == pos_item;
      v = same ? pos_dot : row[j];
      const bool counted = !(catalog && same) && (mask_fn ? v < pos_dot : true);
      return counted ? (catalog ? 1 : (int)a.neg_mult[j]) : 0;
    };
    auto get_c = [&](int j, float& v) {
      const int it = catalog ? j : a.neg_item[j];
      const bool same = it == pos_item;
      const float sv = same ? pos_dot : row[j];
      v = same ? cpos : sv * rq * (catalog ? a.rnorm[j] : a.neg_rc[j]);
      const bool counted = !(catalog && same) && (mask_fn ? v < cpos : true);
      return counted ? (catalog ? 1 : (int)a.neg_mult[j]) : 0;
    };
    radix_select(get_d, N, a.k, hist, sh, out.x, out.y);
    __syncthreads();
    radix_select(get_c, N, a.k, hist, sh, out.z, out.w);
  }
  if (threadIdx.x == 0) a.tau[qi] = out;
}

// ---- combine: one wave per query -----------------------------------------------------------------------
struct CombineArgs {
  const float* tok; const float* table; const float* rnorm;
  const int* counts; const int* qrow; const int* qpos;
  const float* part;       // records of the gradient pass (m, l, sw, counts as the gradient weights used them)
  const float* part_loss;  // records the loss values / statistics are read from (== part in single-pass modes)
  const float* partO;
  float* d_tok; double* blockpart;
  int T, H, nsplit, nsplit_loss, train_head, need_grad, mode; int64_t n_rows;  // nsplit: of `part` / partO; nsplit_loss: of part_loss
  float scale, margin;
  int k_hard, skip_train_head, lse_from_grad;
};

__device__ __forceinline__ double shfl_xor_f64(double v, int o);
constexpr int kCombineRows = 8;  // query rows per workgroup of loss_combine_kernel (two per wave)
constexpr int kValueRows = 256;  // ... of its values-only form (one lane per row)
// VALUES_ONLY: the gradient has been written by the gradient pass itself (LossArgs::d_tok): one LANE per query row merges
// the records and evaluates the row's seven losses + statistics (the half-wave-per-row form spent 38 us on 6.5 + 13 MB).
template <bool VALUES_ONLY>
__global__ __launch_bounds__(256) void loss_combine_kernel(CombineArgs a) {
  // a HALF wave per query row: 32 lanes x 16 bytes cover 128 columns per access (4-byte accesses, one row per wave,
  // ran at 113 us for 260 MB: the kernel is bound by the number of vector-memory instructions)
  __shared__ double sAcc[kCombineRows][BP];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31;
  const int Nq = a.counts[1];
  const int N = (a.mode == XFMR_NEG_CATALOG) ? (int)a.n_rows : a.counts[0];
  const int qi = VALUES_ONLY ? blockIdx.x * kValueRows + (int)threadIdx.x : blockIdx.x * kCombineRows + wid * 2 + half;
  const int H = a.H;
  double acc[BP];
#pragma unroll
  for (int k = 0; k < BP; ++k) acc[k] = 0.0;
  acc[20] = INFINITY; acc[21] = -INFINITY; acc[22] = INFINITY; acc[23] = -INFINITY;
  if (qi < Nq) {
    // merge the split partials (every lane computes the same scalars). Loss VALUES come from part_loss (the
    // logging pass when there is one), everything the gradient is normalised with from the gradient pass.
    const float* r0 = a.part + (int64_t)qi * REC;
    const float pos_dot = r0[R_POSDOT], rq = r0[R_RQ], qq = r0[R_QQ];
    const int pit = a.qpos[qi];
    const float rcpos = a.rnorm[pit];
    const float cpos = pos_dot * rq * rcpos;
    const float sc2 = a.scale * kLog2e;
    const float z2pos = pos_dot * sc2;
    struct Merged { float M, l, cnt_d, cnt_c, sw, nce, hinge, logi, contr, ssum, ssq, smin, smax; };
    auto merge = [&](const float* base, const int ns) {
      Merged g{-INFINITY, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, INFINITY, -INFINITY};
      for (int s = 0; s < ns; ++s) g.M = fmaxf(g.M, base[((int64_t)s * a.T + qi) * REC + R_M]);
      for (int s = 0; s < ns; ++s) {
        const float* r = base + ((int64_t)s * a.T + qi) * REC;
        const float f = exp2f(r[R_M] - g.M);
        g.cnt_d += r[R_CNTD]; g.l += r[R_L] * f; g.nce += r[R_NCE]; g.hinge += r[R_HINGE]; g.logi += r[R_LOGI];
        g.cnt_c += r[R_CNTC]; g.contr += r[R_CONTR]; g.ssum += r[R_SSUM]; g.ssq += r[R_SSQ];
        g.smin = fminf(g.smin, r[R_SMIN]); g.smax = fmaxf(g.smax, r[R_SMAX]);
        g.sw += (a.train_head == XFMR_LOSS_INFONCE) ? r[R_SW] * f : r[R_SW];
      }
      return g;
    };
    const Merged G = merge(a.part, a.nsplit);
    const Merged V = (a.part_loss == a.part) ? G : merge(a.part_loss, a.nsplit_loss);
    {
      const float Mv = a.lse_from_grad ? G.M : V.M, lv = a.lse_from_grad ? G.l : V.l;
      const float ltot_v = lv + exp2f(z2pos - Mv);
      const float inv_dv = 1.f / (V.cnt_d + 1e-9f), inv_cv = 1.f / (V.cnt_c + 1e-9f);
      const float loss_align = 1.f - cpos;
      const float loss_contr = V.contr * inv_cv;
      acc[XFMR_LOSS_ALIGNMENT] = loss_align;
      acc[XFMR_LOSS_ALIGNMENT_CONTRASTIVE] = loss_align + loss_contr;
      acc[XFMR_LOSS_CONTRASTIVE] = loss_contr;
      acc[XFMR_LOSS_INFONCE] = (Mv + log2f(ltot_v)) * kLn2 - a.scale * pos_dot;
      acc[XFMR_LOSS_NCE] = xf_softplus(-pos_dot) + V.nce * inv_dv;
      acc[XFMR_LOSS_PAIRWISE_HINGE] = V.hinge * inv_dv;
      acc[XFMR_LOSS_PAIRWISE_LOGISTIC] = V.logi * inv_dv;
      const int num_neg = (a.k_hard > 0 && a.k_hard < N) ? a.k_hard : N;  // losses.py:387-390
      if (a.skip_train_head) {  // all_heads == 2 (compile-time indices: a runtime index would put acc[] in scratch)
#pragma unroll
        for (int k = 0; k < XFMR_NUM_LOSSES; ++k)
          if (k == a.train_head) acc[k] = 0.0;
      }
      acc[8] = (double)V.cnt_d / ((double)num_neg + 1e-9);  // density term
      acc[9] = pos_dot; acc[10] = (double)pos_dot * pos_dot;
      acc[11] = V.ssum; acc[12] = V.ssq; acc[13] = V.cnt_d; acc[14] = 1.0;
      acc[20] = pos_dot; acc[21] = pos_dot; acc[22] = V.smin; acc[23] = V.smax;
    }
    // gradient-side quantities
    const float M = G.M, sw_ = G.sw;
    const float epos = exp2f(z2pos - M);  // M >= z2pos
    const float ltot = G.l + epos;
    const float inv_d = 1.f / (G.cnt_d + 1e-9f), inv_c = 1.f / (G.cnt_c + 1e-9f);

    if (!VALUES_ONLY && a.need_grad) {
      const int head = a.train_head;
      const int64_t row = a.qrow[qi];
      const float* ep = a.table + (int64_t)pit * H;
      const float* qp = a.tok + row * H;
      float* dst = a.d_tok + row * H;
      const bool cosh = head <= XFMR_LOSS_CONTRASTIVE;
      // first pass for cosine heads: q_hat . dq_hat
      float dotp = 0.f;
      constexpr int NSW = 8;  // 128-column sweeps: H <= 1024
      float4 gk[NSW];         // the row's gradient pieces of this lane
      // the splits' rescaling factors, once per row (InfoNCE: exp2(m_s - M); 1 otherwise; 0 past the plan's split count).
      // Up to 16 splits (every plan's count) the partial rows of a sweep are then loaded EIGHT at a time: one dependent
      // round trip per split and sweep was 13 us of this kernel's 25 on the small steps (round 4, parts compiled out).
      constexpr int FS = 16;
      float fs[FS];
      const int ns = a.nsplit;
#pragma unroll
      for (int u = 0; u < FS; ++u) {
        const int sidx = u < ns ? u : (ns > 0 ? ns - 1 : 0);
        float f = 1.f;
        if (head == XFMR_LOSS_INFONCE) f = exp2f(a.part[((int64_t)sidx * a.T + qi) * REC + R_M] - M);
        fs[u] = u < ns ? f : 0.f;
      }
#pragma unroll
      for (int sw = 0; sw < NSW; ++sw) {
        const int c = sw * 128 + 4 * hl;
        gk[sw] = make_float4(0, 0, 0, 0);
        if (c >= H) continue;
        float4 O = make_float4(0, 0, 0, 0);
        if (head != XFMR_LOSS_ALIGNMENT) {
          if (ns <= FS) {
#pragma unroll
            for (int s0 = 0; s0 < FS; s0 += 8) {
              if (s0 >= ns) break;
              float4 v[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {  // (past the count: the last split's row again, weighted 0)
                const int sidx = s0 + u < ns ? s0 + u : ns - 1;
                v[u] = *reinterpret_cast<const float4*>(&a.partO[((int64_t)sidx * a.T + qi) * H + c]);
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const float f = fs[s0 + u];
                O.x += v[u].x * f; O.y += v[u].y * f; O.z += v[u].z * f; O.w += v[u].w * f;
              }
            }
          } else {
            for (int s = 0; s < ns; ++s) {
              float4 v = *reinterpret_cast<const float4*>(&a.partO[((int64_t)s * a.T + qi) * H + c]);
              float f = 1.f;
              if (head == XFMR_LOSS_INFONCE) f = exp2f(a.part[((int64_t)s * a.T + qi) * REC + R_M] - M);
              O.x += v.x * f; O.y += v.y * f; O.z += v.z * f; O.w += v.w * f;
            }
          }
        }
        const float4 e4 = *reinterpret_cast<const float4*>(ep + c);
        const float Ov[4] = {O.x, O.y, O.z, O.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
        float gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float Ou = Ov[u], e = ev[u];
          float g;
          switch (head) {
            case XFMR_LOSS_INFONCE: g = a.scale * (Ou / ltot - (1.f - epos / ltot) * e); break;
            case XFMR_LOSS_NCE: g = -xf_sigmoid(-pos_dot) * e + Ou * inv_d; break;
            case XFMR_LOSS_PAIRWISE_HINGE:
            case XFMR_LOSS_PAIRWISE_LOGISTIC: g = (Ou - (1.f - a.margin) * sw_ * e) * inv_d; break;
            case XFMR_LOSS_ALIGNMENT: g = -rcpos * e; break;
            case XFMR_LOSS_CONTRASTIVE: g = Ou * inv_c; break;
            default: g = -rcpos * e + Ou * inv_c; break;  // ALIGNMENT_CONTRASTIVE
          }
          gv[u] = g;
        }
        gk[sw] = make_float4(gv[0], gv[1], gv[2], gv[3]);
        if (cosh) {
          const float4 q4 = *reinterpret_cast<const float4*>(qp + c);
          dotp += (gv[0] * q4.x + gv[1] * q4.y + gv[2] * q4.z + gv[3] * q4.w) * rq;
        } else {
          *reinterpret_cast<float4*>(dst + c) = gk[sw];
        }
      }
      if (cosh) {  // (every lane of the half wave runs the same branch: head is uniform)
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dotp += __shfl_xor(dotp, o, 64);
        const bool clamped = sqrtf(qq) < 1e-8f;  // norm clamped: q_hat = q/eps, no projection term
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
          const int c = sw * 128 + 4 * hl;
          if (c >= H) continue;
          const float4 q4 = *reinterpret_cast<const float4*>(qp + c);
          float4 g = gk[sw];
          if (clamped) { g.x *= rq; g.y *= rq; g.z *= rq; g.w *= rq; }
          else {
            g.x = rq * (g.x - q4.x * rq * dotp); g.y = rq * (g.y - q4.y * rq * dotp);
            g.z = rq * (g.z - q4.z * rq * dotp); g.w = rq * (g.w - q4.w * rq * dotp);
          }
          *reinterpret_cast<float4*>(dst + c) = g;
        }
      }
    }
  }
  if (VALUES_ONLY) {  // every lane carries a row: wave reduction (fixed tree), then the four waves through sAcc
#pragma unroll
    for (int k = 0; k < BP; ++k) {
      double v = acc[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double w = shfl_xor_f64(v, o);
        v = (k == 20 || k == 22) ? fmin(v, w) : (k == 21 || k == 23) ? fmax(v, w) : v + w;
      }
      if (lane == 0) sAcc[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < BP) {
      const int k = threadIdx.x;
      double v = sAcc[0][k];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        if (k == 20 || k == 22) v = fmin(v, sAcc[w][k]);
        else if (k == 21 || k == 23) v = fmax(v, sAcc[w][k]);
        else v += sAcc[w][k];
      }
      a.blockpart[(int64_t)k * gridDim.x + blockIdx.x] = v;
    }
    return;
  }
  if (hl == 0) {
#pragma unroll
    for (int k = 0; k < BP; ++k) sAcc[wid * 2 + half][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < BP) {
    const int k = threadIdx.x;
    double v = sAcc[0][k];
#pragma unroll
    for (int w = 1; w < kCombineRows; ++w) {
      if (k == 20 || k == 22) v = fmin(v, sAcc[w][k]);
      else if (k == 21 || k == 23) v = fmax(v, sAcc[w][k]);
      else v += sAcc[w][k];
    }
    a.blockpart[(int64_t)k * gridDim.x + blockIdx.x] = v;  // [BP][blocks]: the final kernel reads coalesced
  }
}

// ---- final: deterministic reduction of the block partials ------------------------------------------------
// 16 waves; wave w reduces quantities k = w, w+16 over the blocks that carried queries (lanes stride the
// blocks, fixed shuffle tree), then one thread finishes the statistics.
__device__ __forceinline__ double shfl_xor_f64(double v, int o) {  // (declared above loss_combine_kernel)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, o, 64);
  hi = __shfl_xor(hi, o, 64);
  return __hiloint2double(hi, lo);
}
// stage 1: block k reduces quantity k over the blocks that carried queries (256 threads, 4 loads in flight each)
__global__ __launch_bounds__(256) void loss_reduce_kernel(const double* blockpart, int nblocks, int rows_per_block,
                                                          const int* counts, double* tot) {
  __shared__ double red[256];
  const int k = blockIdx.x;
  const int Nq = counts[1];
  int used = (Nq + rows_per_block - 1) / rows_per_block;
  if (used > nblocks) used = nblocks;
  const bool is_min = (k == 20 || k == 22), is_max = (k == 21 || k == 23);
  const double id = is_min ? INFINITY : is_max ? -INFINITY : 0.0;
  auto op = [&](double x, double y) { return is_min ? fmin(x, y) : is_max ? fmax(x, y) : x + y; };
  const double* src = blockpart + (int64_t)k * nblocks;
  double v0 = id, v1 = id, v2 = id, v3 = id;
  int b = threadIdx.x;
  for (; b + 768 < used; b += 1024) {
    v0 = op(v0, src[b]); v1 = op(v1, src[b + 256]); v2 = op(v2, src[b + 512]); v3 = op(v3, src[b + 768]);
  }
  for (; b < used; b += 256) v0 = op(v0, src[b]);
  red[threadIdx.x] = op(op(v0, v1), op(v2, v3));
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] = op(red[threadIdx.x], red[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) tot[k] = red[0];
}
// stage 2: one thread turns the 24 totals into the 14 loss values and the statistics
__global__ void loss_final_kernel(const double* tot, const int* counts, int mode, int64_t n_rows, int64_t positions,
                                  float* losses, float* stats) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int Nq = counts[1];
    for (int k = 0; k < XFMR_NUM_LOSSES; ++k) {
      losses[k] = (float)tot[k];
      losses[XFMR_NUM_LOSSES + k] = (float)(tot[k] / ((double)Nq + 1e-9));  // loss/<Class>Mean (trainer.py:263)
    }
    for (int k = 0; k < XFMR_NUM_STATS; ++k) stats[k] = 0.f;
    const double nq = (double)Nq;
    stats[XFMR_STAT_N_VALID] = (float)((mode == XFMR_NEG_CATALOG) ? (int)n_rows : counts[0]);
    stats[XFMR_STAT_N_QUERY] = (float)Nq;
    stats[XFMR_STAT_NEG_DISTINCT] = (float)((mode == XFMR_NEG_CATALOG) ? (int)n_rows : counts[2]);
    // batch/positive_density, batch/attention_density (trainer.py:241-249): on the device, so that the step's logged
    // values need no host arithmetic (and no host sync)
    stats[XFMR_STAT_POS_DENSITY] = (float)(nq / ((double)counts[0] + 1e-9));
    stats[XFMR_STAT_ATTN_DENSITY] = (float)((double)counts[0] / ((double)positions + 1e-9));
    const double nan = __longlong_as_double(0x7ff8000000000000LL);
    stats[XFMR_STAT_NEG_DENSITY] = (float)(Nq > 0 ? tot[8] / nq : nan);
    stats[XFMR_STAT_POS_MEAN] = (float)(Nq > 0 ? tot[9] / nq : nan);
    stats[XFMR_STAT_POS_STD] = (float)(Nq > 1 ? sqrt(fmax(0.0, (tot[10] - tot[9] * tot[9] / nq) / (nq - 1.0))) : nan);
    stats[XFMR_STAT_POS_MIN] = (float)(Nq > 0 ? tot[20] : nan);
    stats[XFMR_STAT_POS_MAX] = (float)(Nq > 0 ? tot[21] : nan);
    const double nn = tot[13];
    stats[XFMR_STAT_NEG_COUNT] = (float)nn;
    stats[XFMR_STAT_NEG_MEAN] = (float)(nn > 0 ? tot[11] / nn : nan);
    stats[XFMR_STAT_NEG_STD] = (float)(nn > 1 ? sqrt(fmax(0.0, (tot[12] - tot[11] * tot[11] / nn) / (nn - 1.0))) : nan);
    stats[XFMR_STAT_NEG_MIN] = (float)(nn > 0 ? tot[22] : nan);
    stats[XFMR_STAT_NEG_MAX] = (float)(nn > 0 ? tot[23] : nan);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
struct Plan {
  int nsplit;
  size_t off_part2, off_negrc, off_tot;
  size_t off_dump, off_tau, off_qinfo; int64_t dump_ld;  // num_hard_negatives only
  size_t off_hist, off_dcnt, off_negmul; int ndblocks;    // distinct-item compaction
  size_t off_counts, off_blockcnt, off_neg, off_qrow, off_qpos, off_part, off_partO, off_block, total;
  int nblocks;
};
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
// `ns_override` (xfmr_loss_cfg.flags, XFMR_LOSS_NSPLIT(n)): a column-split count forced by the caller (parity tests walk
// several plans in one process); 0 = the plan's own choice below.
Plan make_plan(int64_t T, int H, int64_t n_rows, bool hard = false, int ns_override = 0) {
  Plan p;
  const int64_t qblocks = (T + QB - 1) / QB;
  const int64_t cols = T > n_rows ? T : n_rows;
  const int64_t tiles = (cols + BN - 1) / BN;
  int64_t ns = (1024 + qblocks / 2) / qblocks;  // ~1024 workgroups: two full rounds at 2 workgroups/CU
  if (ns > 16) ns = 16;
  // 32 ... 63 query blocks (batch 32 at L = 200 is 50): 16 splits leave each workgroup 4 of the 61 column tiles behind a
  // query prologue worth ~10, and the combine kernel 16 partial records per query: 8 splits 0.790 ms/step, 16 0.830, 4
  // 0.794 (round 3, one box). At 100 blocks (batch 64) the formula's 10 measured better than 8 (1.000 against 1.013);
  // fewer than 32 blocks (the reference's default model: 8) keep the finer split.
  if (qblocks >= 32 && qblocks < 64 && ns > 8) ns = 8;
  if (ns < 2) ns = 2;  // measured at 800 query blocks (B = 512): 2 splits 95.5k seq/s, 1 split 94.1k
  // >= 512 query blocks: the gradient pass runs ONE split there (run_loss), so this count is the logging pass's alone. At
  // three workgroups per CU (768 slots) 800 blocks x 2 splits are 2.08 rounds -- the third round nearly empty; finer
  // pieces fill it: isolated 456 us (2 splits), 427 (3), 422 (4), 414 (5), 425 (6), 428 (8); 525 with one. In the step the
  // pass is hidden underneath the backward either way (3.32-3.36 ms for 2 ... 6 splits).
  if (ns < 4 && qblocks >= 512) ns = 4;
  static const int ns_env = [] { const char* e = getenv("XFMR_LOSS_NSPLIT"); return e ? atoi(e) : 0; }();  // tuning experiments
  if (ns_env > 0) ns = ns_env;
  if (ns_override > 0) ns = ns_override;
  if (ns > tiles) ns = tiles;
  if (ns < 1) ns = 1;
  p.nsplit = (int)ns;
  p.nblocks = (int)((T + kCombineRows - 1) / kCombineRows);
  size_t o = 0;
  p.off_counts = o; o += 256;
  p.off_blockcnt = o; o += up256((size_t)((T + PREP - 1) / PREP) * sizeof(int2));
  p.off_neg = o; o += up256((size_t)T * 4);
  p.off_negrc = o; o += up256((size_t)T * 4);
  p.off_negmul = o; o += up256((size_t)T * 4);
  p.ndblocks = (int)((n_rows + PREP - 1) / PREP);
  p.off_hist = o; o += up256((size_t)n_rows * 4);
  p.off_dcnt = o; o += up256((size_t)p.ndblocks * 4);
  p.off_qrow = o; o += up256((size_t)T * 4);
  p.off_qpos = o; o += up256((size_t)T * 4);
  p.off_part = o; o += up256((size_t)ns * T * REC * 4);
  p.off_part2 = o; o += up256((size_t)ns * T * REC * 4);
  p.off_partO = o; o += up256((size_t)ns * T * H * 4);
  p.off_block = o; o += up256((size_t)p.nblocks * BP * 8);
  p.off_tot = o; o += 256;
  p.off_dump = p.off_tau = p.off_qinfo = 0;
  p.dump_ld = ((cols + BN - 1) / BN) * BN;  // whole tiles: the dump writes 16-byte runs
  if (hard) {
    p.off_tau = o; o += up256((size_t)T * sizeof(float4));
    p.off_qinfo = o; o += up256((size_t)T * sizeof(float2));
    p.off_dump = o; o += up256((size_t)T * (size_t)p.dump_ld * sizeof(float));
  }
  p.total = o;
  return p;
}

template <class P, int H>
void launch_main(const LossArgs& a, bool all, dim3 grid, hipStream_t st) {
  grid.z = (a.need_grad && !a.dump) ? loss_main_hparts<P, H>() : 1;  // dQ column parts (values-only launches: one)
  if (all) hipLaunchKernelGGL((loss_main_kernel<P, H, true>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((loss_main_kernel<P, H, false>), grid, dim3(256), 0, st, a);
}
// H: the rows' real width (any multiple of 32) -> the next instantiated template width (see loss_main_kernel)
template <class P>
int launch_main_h(const LossArgs& a, int H, bool all, dim3 grid, hipStream_t st) {
  if (H <= 0 || (H & 31)) return XFMR_EUNSUPPORTED;
  constexpr bool f32 = sizeof(typename P::elem) == 4;
  if (H <= 64) launch_main<P, 64>(a, all, grid, st);
  else if (H <= 128) launch_main<P, 128>(a, all, grid, st);
  else if (H <= 256) launch_main<P, 256>(a, all, grid, st);
  else if (H <= 384) launch_main<P, 384>(a, all, grid, st);
  else if (H <= 512) launch_main<P, 512>(a, all, grid, st);
  else if (f32) return XFMR_EUNSUPPORTED;  // (the fp32 tile image of 64 x 768 does not fit LDS)
  else if (H <= 768) launch_main<PrecBF16, 768>(a, all, grid, st);
  else if (H <= 1024) launch_main<PrecBF16, 1024>(a, all, grid, st);
  else return XFMR_EUNSUPPORTED;
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int launch_dma_h(const LossArgs& a, const void* tbf, int H, int head, dim3 grid, hipStream_t st) {
  switch (H) {
    case 64: return xf_launch_loss_dma_64(a, tbf, head, grid, st);
    case 128: return xf_launch_loss_dma_128(a, tbf, head, grid, st);
    case 256: return xf_launch_loss_dma_256(a, tbf, head, grid, st);
    case 384: return xf_launch_loss_dma_384(a, tbf, head, grid, st);
    default: return XFMR_EUNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int xf_loss_finalize(const double* blockpart, int nblocks, int rows_per_block, const int* counts, int mode,
                     int64_t n_rows, int64_t positions, float* losses, float* stats, double* tot, hipStream_t st) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(BP), dim3(256), 0, st, blockpart, nblocks, rows_per_block, counts, tot);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const double*)tot, counts, mode, n_rows, positions,
                     losses, stats);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

size_t xfmr_sampled_loss_workspace(int64_t positions, int32_t H, int64_t n_rows) {
  if (positions <= 0 || H <= 0) return 0;
  return make_plan(positions, H, n_rows).total;
}
size_t xfmr_sampled_loss_workspace_cfg(const xfmr_loss_cfg* cfg, int64_t positions, int32_t H, int64_t n_rows) {
  if (!cfg || positions <= 0 || H <= 0) return 0;
  return make_plan(positions, H, n_rows, cfg->num_hard_negatives > 0, (int)XFMR_LOSS_NSPLIT_OF(cfg->flags)).total;
}

static int run_loss(const xfmr_loss_cfg* cfg, const float* tok, const float* table, const float* table_rnorm,
                    const void* table_bf16, int64_t n_rows, int T, int32_t H, float* losses, float* stats, float* d_tok, unsigned char* ws,
                    const Plan& p, hipStream_t st) {
  int* counts = (int*)(ws + p.off_counts);
  int* neg_item = (int*)(ws + p.off_neg);
  int* qrow = (int*)(ws + p.off_qrow);
  int* qpos = (int*)(ws + p.off_qpos);
  LossArgs a{};
  a.tok = tok; a.table = table; a.rnorm = table_rnorm; a.n_rows = n_rows; a.counts = counts;
  a.neg_item = cfg->mode == XFMR_NEG_SHARED ? neg_item : nullptr;
  a.neg_rc = (const float*)(ws + p.off_negrc);
  a.neg_mult = cfg->mode == XFMR_NEG_SHARED ? (const float*)(ws + p.off_negmul) : nullptr;
  a.qrow = qrow; a.qpos = qpos; a.part = (float*)(ws + p.off_part); a.partO = (float*)(ws + p.off_partO);
  a.T = T; a.H = H; a.nsplit = p.nsplit; a.train_head = cfg->train_head; a.mask_fn = cfg->mask_false_negatives;
  a.mode = cfg->mode; a.need_grad = d_tok != nullptr; a.scale = cfg->scale; a.margin = cfg->margin;
  dim3 grid((unsigned)((T + QB - 1) / QB), p.nsplit);
  // Gradient pass of the bf16 production path with ONE column split when the query blocks alone fill the chip (>= two
  // workgroups per CU): the kernel then finishes its rows itself (LossArgs::d_tok) -- no (split, T, H) partial dQ written
  // and re-read (105 + 105 MB at the benchmark shape), no gradient work in the combine kernel. Measured at 800 query
  // blocks: the pass itself takes the same time with 1, 2 or 3 splits (387 us). XFMR_LOSS_NSPLIT_GRAD overrides (experiments).
  int ns_grad = p.nsplit;
  // (round 4: from 257 blocks on. n splits of b blocks take ceil(n b / 512) rounds of 1/n of the columns each: with more than
  //  256 blocks two splits are two rounds of half a walk -- one split's time -- plus the partial dQ and the combine launch.
  //  Measured on MovieLens-like batches of 512 in the packed layout, ~390 blocks: 2.00-2.01 against 2.03 ms per step.)
  if (grid.x >= 512 || (H <= 128 && grid.x > 256)) ns_grad = 1;  // (H > 128: one workgroup per CU, planned by rounds below)
  static const int ns_grad_env = [] { const char* e = getenv("XFMR_LOSS_NSPLIT_GRAD"); return e ? atoi(e) : 0; }();
  if (ns_grad_env > 0) ns_grad = ns_grad_env;
  const int ns_grad_flag = (int)XFMR_LOSS_NSPLIT_GRAD_OF(cfg->flags);  // (tests: several plans in one process)
  if (ns_grad_flag > 0) ns_grad = ns_grad_flag;
  if (ns_grad < 1) ns_grad = 1;
  if (ns_grad > p.nsplit) ns_grad = p.nsplit;  // (the workspace is carved for p.nsplit)
  int ns_part = p.nsplit;  // split count the records in a.part (and partO) are written with
  bool fused_finish = false;
  int rc;
  // measurement events (xfmr_loss_cfg): profile_grad around the gradient pass -- or around the call's single main kernel
  // when it runs the generic kernel --, profile_log around the values-only logging pass of the bf16 production path
  hipEvent_t ev0 = (hipEvent_t)cfg->profile_grad[0], ev1 = (hipEvent_t)cfg->profile_grad[1];
  hipEvent_t lev0 = (hipEvent_t)cfg->profile_log[0], lev1 = (hipEvent_t)cfg->profile_log[1];
  const float* part_loss = nullptr;  // records the loss VALUES are read from (null: the same as the gradient's)
  bool lse_from_grad = false;        // InfoNCE value from the gradient pass's records (logging pass ran without it)
  const bool hard = cfg->num_hard_negatives > 0;
  if (hard) {
    // top-k hard negatives (losses.py:295-330), generic kernel for both precisions: (1) the same kernel dumps its
    // logits (bit-identical to what the loss pass will see), (2) per-row thresholds by radix select, (3) the loss
    // pass with every counted negative weighted by its top-k weight.
    LossArgs d = a;
    d.dump = (float*)(ws + p.off_dump); d.dump_ld = p.dump_ld; d.qinfo = (float2*)(ws + p.off_qinfo); d.need_grad = 0;
    if (cfg->precision == XFMR_PREC_BF16) rc = launch_main_h<PrecBF16>(d, H, false, grid, st);
    else if (cfg->precision == XFMR_PREC_F32) rc = launch_main_h<PrecF32>(d, H, false, grid, st);
    else rc = XFMR_EINVAL;
    if (rc) return rc;
    SelectArgs s{};
    s.dump = d.dump; s.dump_ld = p.dump_ld; s.qinfo = d.qinfo; s.counts = counts; s.neg_item = neg_item;
    s.neg_rc = a.neg_rc; s.neg_mult = a.neg_mult; s.rnorm = table_rnorm; s.qpos = qpos; s.tau = (float4*)(ws + p.off_tau);
    s.mode = cfg->mode; s.mask_fn = cfg->mask_false_negatives; s.k = cfg->num_hard_negatives; s.n_rows = n_rows;
    hipLaunchKernelGGL(hard_select_kernel, dim3((unsigned)T), dim3(256), 0, st, s);
    XF_LAUNCH_CHECK();
    a.tau = (const float4*)(ws + p.off_tau);
  }
  // widths the LDS-DMA kernel is instantiated for; its gather addresses rows by a 32-bit byte offset from the table base
  const bool dma_width = (H == 64 || H == 128 || H == 256 || H == 384) && (uint64_t)n_rows * (uint64_t)H * 2 < (1ull << 32);
  if (!hard && cfg->precision == XFMR_PREC_BF16 && table_bf16 && dma_width) {
    // bf16 production path: the gradient pass of the train head (skipped for AlignmentLoss, whose gradient has
    // no negative term) and, when every head is wanted or no gradient is, the values-only logging pass.
    const bool all = cfg->all_heads != 0;
    const bool grad_pass = a.need_grad && cfg->train_head != XFMR_LOSS_ALIGNMENT;
    // unmasked InfoNCE with the logging pass in the same call: logging first, then the gradient pass with the row
    // maximum pinned from the logging records (lean epilogue, see HEAD_INFONCE_PINNED)
    const bool pinned = grad_pass && cfg->all_heads == 1 && cfg->train_head == XFMR_LOSS_INFONCE &&
                        !cfg->mask_false_negatives;
    // the online-maximum InfoNCE at H > 128 runs as dQ column parts (loss_dma.inc): it keeps the split form
    const bool col_parts = (H > 128 && cfg->train_head == XFMR_LOSS_INFONCE && !cfg->mask_false_negatives && !pinned) ||
                           (H == 256 && grad_pass && !pinned && xf_loss_dma_256_hparts(a, cfg->train_head) > 1);
    // H > 128 below 512 query blocks: these gradient kernels run ONE workgroup per CU (loss_dma.inc), all of them equally
    // long, so the pass takes ceil(blocks * n / CUs) rounds of 1 / n of the columns each. The fewest splits among the best
    // such n: config 5 (256 blocks on 256 CUs) runs 1 split -- one full round, the rows finished in the kernel, no 134 MB
    // of partial dQ written and re-read (6.62 -> 6.51 ms/step); config 4 (100 blocks) 5 splits instead of 10 (two rounds
    // of 250: 3.06 -> 2.99; 3 splits = 1.17 rounds measured 3.30).
    if (H > 128 && grid.x < 512 && ns_grad_env <= 0 && ns_grad_flag <= 0) {
      static const int cus = [] {
        int dev = 0, cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        return cu > 0 ? cu : 256;
      }();
      int best = p.nsplit;
      double best_cost = 1e30;
      for (int n = 1; n <= p.nsplit; ++n) {
        // (+ 0.01 per split: every split writes and the combine kernel re-reads a (queries x H) fp32 partial dQ -- at 8 query
        //  blocks (the reference's default model: 32 x 32 tokens, H = 384) 8 splits measured 0.303 ms per step against 0.313
        //  for 16 and 0.306 for 4; configs 4 / 5 keep their 5 / 1)
        const double cost = (double)(((int64_t)grid.x * n + cus - 1) / cus) / n + 0.01 * n;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
      }
      ns_grad = best;
    }
    if (!grad_pass || col_parts) {  // column parts keep the split form (no in-kernel finish); overrides still apply
      ns_grad = p.nsplit;
      if (col_parts && ns_grad_env > 0) ns_grad = ns_grad_env < p.nsplit ? ns_grad_env : p.nsplit;
      if (col_parts && ns_grad_flag > 0) ns_grad = ns_grad_flag < p.nsplit ? ns_grad_flag : p.nsplit;
    }
    dim3 ggrid(grid.x, (unsigned)ns_grad);
    if (grad_pass) {
      a.nsplit = ns_part = ns_grad;
      if (ns_grad == 1 && !col_parts) { a.d_tok = d_tok; fused_finish = true; }
    }
    if (pinned) {
      LossArgs b = a;
      b.nsplit = p.nsplit; b.d_tok = nullptr;
      b.part = (float*)(ws + p.off_part2);
      b.need_grad = 0;
      if (lev0 && hipEventRecord(lev0, st) != hipSuccess) return XFMR_EHIP;
      rc = launch_dma_h(b, table_bf16, H, -2, grid, st);  // (the InfoNCE value comes from the gradient records)
      if (rc) return rc;
      if (lev1 && hipEventRecord(lev1, st) != hipSuccess) return XFMR_EHIP;
      part_loss = b.part;
      lse_from_grad = true;
      a.pin_part = b.part;
      a.pin_nsplit = p.nsplit;
      if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return XFMR_EHIP;
      rc = launch_dma_h(a, table_bf16, H, cfg->train_head, ggrid, st);
      if (rc) return rc;
      if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return XFMR_EHIP;
    } else {
    if (grad_pass) {
      if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return XFMR_EHIP;
      rc = launch_dma_h(a, table_bf16, H, cfg->train_head, ggrid, st);
      if (rc) return rc;
      if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return XFMR_EHIP;
    }
    if (all || !grad_pass) {
      LossArgs b = a;
      b.nsplit = p.nsplit; b.d_tok = nullptr;
      if (grad_pass) b.part = (float*)(ws + p.off_part2);
      b.need_grad = 0;
      // all_heads == 2: the train head's value is not wanted from this call (the caller has it from the gradient
      // pass of the same step): for InfoNCE that drops the log-sum-exp from the per-logit work
      // ... and likewise when the gradient pass of THIS call is the InfoNCE one: the combine kernel then takes the
      // head's value from the gradient pass's records
      const bool lse_elsewhere = cfg->train_head == XFMR_LOSS_INFONCE && (grad_pass || cfg->all_heads == 2);
      lse_from_grad = lse_elsewhere && grad_pass;
      const int code = lse_elsewhere ? -2 : -1;
      if (lev0 && hipEventRecord(lev0, st) != hipSuccess) return XFMR_EHIP;
      rc = launch_dma_h(b, table_bf16, H, code, grid, st);
      if (rc) return rc;
      if (lev1 && hipEventRecord(lev1, st) != hipSuccess) return XFMR_EHIP;
      if (grad_pass) part_loss = b.part;
    }
    }
  } else {
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return XFMR_EHIP;
    if (cfg->precision == XFMR_PREC_BF16) rc = launch_main_h<PrecBF16>(a, H, cfg->all_heads != 0, grid, st);
    else if (cfg->precision == XFMR_PREC_F32) rc = launch_main_h<PrecF32>(a, H, cfg->all_heads != 0, grid, st);
    else rc = XFMR_EINVAL;
    if (rc) return rc;
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return XFMR_EHIP;
  }

  CombineArgs c{};
  c.tok = tok; c.table = table; c.rnorm = table_rnorm; c.counts = counts; c.qrow = qrow; c.qpos = qpos;
  c.part = a.part; c.part_loss = part_loss ? part_loss : a.part; c.partO = a.partO; c.d_tok = d_tok; c.blockpart = (double*)(ws + p.off_block);
  c.T = T; c.H = H; c.train_head = cfg->train_head;
  // split counts of the two record sets: `part` (and partO) were written by the launch that used a.nsplit; part_loss,
  // when it is another buffer, by a logging pass with p.nsplit
  c.nsplit = ns_part; c.nsplit_loss = part_loss ? p.nsplit : ns_part;
  c.need_grad = (d_tok != nullptr && !fused_finish) ? 1 : 0;  // fused_finish: the gradient pass wrote d_tok itself
  c.mode = cfg->mode; c.n_rows = n_rows; c.scale = cfg->scale; c.margin = cfg->margin;
  c.k_hard = cfg->num_hard_negatives;
  c.skip_train_head = (cfg->all_heads == 2 && d_tok == nullptr) ? 1 : 0;
  c.lse_from_grad = lse_from_grad ? 1 : 0;
  if (!c.need_grad) {  // values only: one lane per row
    const int nb = (T + kValueRows - 1) / kValueRows;
    hipLaunchKernelGGL(loss_combine_kernel<true>, dim3(nb), dim3(256), 0, st, c);
    XF_LAUNCH_CHECK();
    return xf_loss_finalize(c.blockpart, nb, kValueRows, counts, cfg->mode, n_rows, cfg->padded_positions > 0 ? cfg->padded_positions : (int64_t)T, losses, stats,
                            (double*)(ws + p.off_tot), st);
  }
  hipLaunchKernelGGL(loss_combine_kernel<false>, dim3(p.nblocks), dim3(256), 0, st, c);
  XF_LAUNCH_CHECK();
  return xf_loss_finalize(c.blockpart, p.nblocks, kCombineRows, counts, cfg->mode, n_rows, cfg->padded_positions > 0 ? cfg->padded_positions : (int64_t)T, losses, stats,
                          (double*)(ws + p.off_tot), st);
}

// histogram over the catalogue -> distinct negative items, inverse norms, multiplicities, counts[2]
static int launch_distinct(unsigned char* ws, const Plan& p, int64_t n_rows, const float* table_rnorm, hipStream_t st) {
  const int* hist = (const int*)(ws + p.off_hist);
  int* dcnt = (int*)(ws + p.off_dcnt);
  hipLaunchKernelGGL(distinct_count_kernel, dim3(p.ndblocks), dim3(PREP), 0, st, hist, n_rows, dcnt);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL(distinct_write_kernel, dim3(p.ndblocks), dim3(PREP), 0, st, hist, n_rows, (const int*)dcnt,
                     table_rnorm, (int*)(ws + p.off_counts), (int*)(ws + p.off_neg), (float*)(ws + p.off_negrc),
                     (float*)(ws + p.off_negmul));
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

static int check_loss_args(const xfmr_loss_cfg* cfg, const float* tok, const float* table, const float* rnorm,
                           float* losses, float* stats, void* workspace, float* d_tok, int64_t rows, int64_t n_rows) {
  if (!cfg || !tok || !table || !rnorm || !losses || !stats || !workspace) return XFMR_EINVAL;
  if (rows <= 0 || n_rows <= 0 || rows > (1 << 30) || n_rows > (1 << 30)) return XFMR_EINVAL;
  if (cfg->train_head < 0 || cfg->train_head >= XFMR_NUM_LOSSES || cfg->num_hard_negatives < 0) return XFMR_EINVAL;
  if (!xf_aligned16(tok) || !xf_aligned16(table) || !xf_aligned16(workspace) || (d_tok && !xf_aligned16(d_tok)))
    return XFMR_EALIGN;
  return XFMR_OK;
}

// The part of xfmr_sampled_loss that depends only on the key mask and the index tensors (not on the token embeddings): the
// compacted query list, the multiplicity histogram of the shared negatives and their distinct-item list, in `workspace`.
int xfmr_sampled_loss_prepare(const xfmr_loss_cfg* cfg, const uint8_t* key_mask, const int64_t* pos_idx,
                              const int64_t* neg_idx, const float* table_rnorm, int64_t n_rows, int64_t positions,
                              int32_t H, void* workspace, size_t workspace_bytes, void* stream) {
  if (!cfg || !key_mask || !pos_idx || !table_rnorm || !workspace) return XFMR_EINVAL;
  if (positions <= 0 || n_rows <= 0 || positions > (1 << 30) || n_rows > (1 << 30) || cfg->num_hard_negatives < 0)
    return XFMR_EINVAL;
  if (cfg->mode == XFMR_NEG_SHARED && !neg_idx) return XFMR_EINVAL;
  if (!xf_aligned16(workspace)) return XFMR_EALIGN;
  const Plan p = make_plan(positions, H, n_rows, cfg->num_hard_negatives > 0, (int)XFMR_LOSS_NSPLIT_OF(cfg->flags));
  if (workspace_bytes < p.total) return XFMR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  const int T = (int)positions;
  const int nprep = (T + PREP - 1) / PREP;
  int2* blockcnt = (int2*)(ws + p.off_blockcnt);
  hipLaunchKernelGGL(prepare_count_kernel, dim3(nprep), dim3(PREP), 0, st, key_mask, pos_idx, T, blockcnt);
  XF_LAUNCH_CHECK();
  const bool shared = cfg->mode == XFMR_NEG_SHARED;
  int* hist = (int*)(ws + p.off_hist);
  if (shared && xf_zero_async(hist, (size_t)n_rows * sizeof(int), st) != hipSuccess) return XFMR_EHIP;
  hipLaunchKernelGGL(prepare_write_kernel, dim3(nprep), dim3(PREP), 0, st, key_mask, pos_idx,
                     shared ? neg_idx : (const int64_t*)nullptr, T, n_rows, (const int2*)blockcnt,
                     (int*)(ws + p.off_counts), hist, (int*)(ws + p.off_qrow), (int*)(ws + p.off_qpos));
  XF_LAUNCH_CHECK();
  if (shared) {
    if (int rc = launch_distinct(ws, p, n_rows, table_rnorm, st)) return rc;
  }
  return XFMR_OK;
}

// ... and the rest, on a workspace xfmr_sampled_loss_prepare filled for the SAME cfg / key mask / index tensors / sizes.
int xfmr_sampled_loss_prepared(const xfmr_loss_cfg* cfg, const float* tok, const uint8_t* key_mask, const int64_t* pos_idx,
                               const int64_t* neg_idx, const float* table, const float* table_rnorm, const void* table_bf16,
                               int64_t n_rows, int64_t positions, int32_t H, float* losses, float* stats, float* d_tok,
                               void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_loss_args(cfg, tok, table, table_rnorm, losses, stats, workspace, d_tok, positions, n_rows))
    return rc;
  if (!key_mask || !pos_idx) return XFMR_EINVAL;
  if (cfg->mode == XFMR_NEG_SHARED && !neg_idx) return XFMR_EINVAL;
  const Plan p = make_plan(positions, H, n_rows, cfg->num_hard_negatives > 0, (int)XFMR_LOSS_NSPLIT_OF(cfg->flags));
  if (workspace_bytes < p.total) return XFMR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool zeroed = (cfg->flags & XFMR_LOSS_DTOK_ZEROED) != 0;  // the caller zeroed d_tok itself
  if (d_tok && !zeroed && xf_zero_async(d_tok, (size_t)positions * H * sizeof(float), st) != hipSuccess) return XFMR_EHIP;
  if (table_bf16 && !xf_aligned16(table_bf16)) return XFMR_EALIGN;
  return run_loss(cfg, tok, table, table_rnorm, table_bf16, n_rows, (int)positions, H, losses, stats, d_tok,
                  (unsigned char*)workspace, p, st);
}

int xfmr_sampled_loss(const xfmr_loss_cfg* cfg, const float* tok, const uint8_t* key_mask, const int64_t* pos_idx,
                      const int64_t* neg_idx, const float* table, const float* table_rnorm, const void* table_bf16,
                      int64_t n_rows, int64_t positions, int32_t H, float* losses, float* stats, float* d_tok, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (int rc = check_loss_args(cfg, tok, table, table_rnorm, losses, stats, workspace, d_tok, positions, n_rows))
    return rc;
  if (int rc = xfmr_sampled_loss_prepare(cfg, key_mask, pos_idx, neg_idx, table_rnorm, n_rows, positions, H, workspace,
                                         workspace_bytes, stream))
    return rc;
  return xfmr_sampled_loss_prepared(cfg, tok, key_mask, pos_idx, neg_idx, table, table_rnorm, table_bf16, n_rows, positions,
                                    H, losses, stats, d_tok, workspace, workspace_bytes, stream);
}

size_t xfmr_sampled_loss_lists_workspace(int64_t n_query, int64_t n_neg, int32_t H, int64_t n_rows) {
  const int64_t rows = n_query > n_neg ? n_query : n_neg;
  if (rows <= 0 || H <= 0) return 0;
  return make_plan(rows, H, n_rows).total;
}
size_t xfmr_sampled_loss_lists_workspace_cfg(const xfmr_loss_cfg* cfg, int64_t n_query, int64_t n_neg, int32_t H,
                                             int64_t n_rows) {
  const int64_t rows = n_query > n_neg ? n_query : n_neg;
  if (!cfg || rows <= 0 || H <= 0) return 0;
  return make_plan(rows, H, n_rows, cfg->num_hard_negatives > 0, (int)XFMR_LOSS_NSPLIT_OF(cfg->flags)).total;
}

int xfmr_sampled_loss_lists(const xfmr_loss_cfg* cfg, const float* query, const int64_t* pos_items,
                            const int64_t* neg_items, int64_t n_query, int64_t n_neg, const float* table,
                            const float* table_rnorm, const void* table_bf16, int64_t n_rows, int32_t H,
                            float* losses, float* stats, float* d_query, void* workspace, size_t workspace_bytes, void* stream) {
  const int64_t rows = n_query > n_neg ? n_query : n_neg;
  if (int rc = check_loss_args(cfg, query, table, table_rnorm, losses, stats, workspace, d_query, rows, n_rows))
    return rc;
  if (!pos_items || n_query <= 0) return XFMR_EINVAL;
  if (cfg->mode == XFMR_NEG_SHARED && (!neg_items || n_neg <= 0)) return XFMR_EINVAL;
  const Plan p = make_plan(rows, H, n_rows, cfg->num_hard_negatives > 0, (int)XFMR_LOSS_NSPLIT_OF(cfg->flags));
  if (workspace_bytes < p.total) return XFMR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  const int T = (int)rows;
  const bool shared = cfg->mode == XFMR_NEG_SHARED;
  int* hist = (int*)(ws + p.off_hist);
  if (shared && xf_zero_async(hist, (size_t)n_rows * sizeof(int), st) != hipSuccess) return XFMR_EHIP;
  hipLaunchKernelGGL(prepare_lists_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, pos_items,
                     shared ? neg_items : (const int64_t*)nullptr, (int)n_query, (int)(shared ? n_neg : 0), n_rows,
                     (int*)(ws + p.off_counts), hist, (int*)(ws + p.off_qrow), (int*)(ws + p.off_qpos));
  XF_LAUNCH_CHECK();
  if (shared) {
    if (int rc = launch_distinct(ws, p, n_rows, table_rnorm, st)) return rc;
  }
  if (table_bf16 && !xf_aligned16(table_bf16)) return XFMR_EALIGN;
  return run_loss(cfg, query, table, table_rnorm, table_bf16, n_rows, T, H, losses, stats, d_query, ws, p, st);
}

int xfmr_table_prepare(const float* table, float* table_rnorm, void* table_bf16, int64_t n_rows, int32_t H,
                       void* stream) {
  if (!table || n_rows <= 0 || H <= 0 || (!table_rnorm && !table_bf16)) return XFMR_EINVAL;
  hipLaunchKernelGGL(table_prepare_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     table, table_rnorm, (__bf16*)table_bf16, n_rows, H);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

}  // extern "C"
