// Dense-candidate form of EmbedLoss.forward (xfmr_rec/losses.py:128-155): query (N,H), candidates (N,C,H) as
// tensors, optional explicit target -- the reference's own calling convention, for the sizes at which that tensor
// is reasonable (C <= 8192). The training path never builds it (see loss.hip); this kernel exists so that the
// whole EmbedLoss contract -- target_position first / diagonal / explicit target, false-negative masking,
// num_hard_negatives, all seven heads + LogitsStatistics, dL/dquery -- is available behind the same C ABI.
//
// One workgroup per row, everything fp32 (vector FMA: the kernel is bound by reading the candidates, twice):
//   1. logits: 16-lane groups take one candidate each (16-byte pieces, shuffle reduction): dot and cosine
//   2. negative masks (losses.py:263-293) and, when num_hard_negatives is set, the top-k restriction
//      (losses.py:295-330) as a per-row threshold on the masked logits: a 32-step bitwise search for the k-th
//      largest key. Elements equal to the threshold share the remaining weight (k - #greater) / #equal: the
//      loss is a function of the multiset of selected logits, so this equals any tie-breaking torch.topk makes
//   3. the row's seven loss values + statistics, written as one fp64 record per row (deterministic totals
//      by the reduce / final kernels shared with the fused path)
//   4. dL(train_head)/dq = sum_c w_c cand_c (+ positive / cosine terms), a second pass over the candidates.
#include "internal.h"
#include "loss_common.h"

namespace {

constexpr int DENSE_MAX_C = 8192;
constexpr int DENSE_MAX_H = 1024;  // gradient: four columns per thread

struct DenseArgs {
  const float* q; const float* cand; const int64_t* target;
  int target_mode;  // 0 first, 1 diagonal, 2 explicit
  int64_t N; int C; int H;
  int train_head, all_heads, mask_fn, k_hard;
  float scale, margin;
  float* d_query; double* blockpart;
  float* d_cand;  // (N, C, H) or null: dL(train_head)/dcandidates (losses.py:128-155 is differentiable in them too)
};

__device__ __forceinline__ unsigned sort_key(float f) {  // monotone: a < b  <=>  key(a) < key(b)
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

struct BlockRed {
  float* red;  // [8]
  __device__ __forceinline__ float sum(float v) {
    v = xf_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  }
  __device__ __forceinline__ float max(float v) {
    v = xf_wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
  __device__ __forceinline__ float min(float v) { return -max(-v); }
};

// weight of every element under the top-k restriction: 1 above the threshold key, rho at it, 0 below
struct TopK {
  unsigned key; float rho; bool on;
  __device__ __forceinline__ float weight(float v) const {
    if (!on) return 1.f;
    const unsigned k = sort_key(v);
    return k > key ? 1.f : (k == key ? rho : 0.f);
  }
};

// counted(c) says whether candidate c passed the false-negative mask; vals = the logits the head ranks by
template <class Counted>
__device__ TopK select_topk(const float* vals, int C, int k, Counted counted, BlockRed& br) {
  TopK t{0u, 1.f, false};
  if (k <= 0 || k >= C) return t;  // losses.py:312-316
  float n = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) n += counted(c) ? 1.f : 0.f;
  n = br.sum(n);
  if (n <= (float)k) return t;  // fewer counted negatives than k: all of them stay
  unsigned T = 0;
  for (int bit = 31; bit >= 0; --bit) {
    const unsigned cand = T | (1u << bit);
    float cnt = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) cnt += (counted(c) && sort_key(vals[c]) >= cand) ? 1.f : 0.f;
    if (br.sum(cnt) >= (float)k) T = cand;
  }
  float gt = 0.f, eq = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    if (!counted(c)) continue;
    const unsigned key = sort_key(vals[c]);
    gt += key > T ? 1.f : 0.f;
    eq += key == T ? 1.f : 0.f;
  }
  gt = br.sum(gt);
  eq = br.sum(eq);
  t.key = T; t.rho = ((float)k - gt) / eq; t.on = true;
  return t;
}

__global__ __launch_bounds__(256) void dense_loss_kernel(DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C, H = a.H;
  float* sQ = smem;              // [H]
  float* sDot = sQ + H;          // [C]
  float* sCos = sDot + C;        // [C]
  float* sRc = sCos + C;         // [C]
  float* sW = sRc + C;           // [C] gradient weight of the train head
  float* sRed = sW + C;          // [8]
  BlockRed br{sRed};
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const float* q = a.q + row * H;
  const float* cand = a.cand + row * (int64_t)C * H;

  float qq = 0.f;
  for (int h = tid; h < H; h += 256) {
    const float v = q[h];
    sQ[h] = v;
    qq += v * v;
  }
  qq = br.sum(qq);  // (also publishes sQ)
  const float rq = 1.f / fmaxf(sqrtf(qq), 1e-8f);

  // ---- 1. logits ----------------------------------------------------------------------------------------
  {
    const int g = tid >> 4, j = tid & 15;
    for (int c = g; c < C; c += 16) {
      const float* e = cand + (int64_t)c * H;
      float dot = 0.f, cc = 0.f;
      for (int h = 4 * j; h < H; h += 64) {
        const float4 x = *reinterpret_cast<const float4*>(e + h);
        const float4 y = *reinterpret_cast<const float4*>(sQ + h);
        dot = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, dot))));
        cc = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, cc))));
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        dot += __shfl_xor(dot, o, 64);
        cc += __shfl_xor(cc, o, 64);
      }
      if (j == 0) {
        const float rc = 1.f / fmaxf(sqrtf(cc), 1e-8f);
        sDot[c] = dot;
        sRc[c] = rc;
        sCos[c] = dot * rq * rc;
      }
    }
  }
  __syncthreads();

  // ---- 2. target, masks, top-k thresholds ---------------------------------------------------------------
  int tgt = 0;
  if (a.target_mode == 1) tgt = (int)row;
  else if (a.target_mode == 2) tgt = (int)a.target[row];
  tgt = min(max(tgt, 0), C - 1);  // the reference's gather would raise on an out-of-range target
  const float pos_dot = sDot[tgt], cpos = sCos[tgt], rcpos = sRc[tgt];
  const bool mask_fn = a.mask_fn != 0;
  auto counted_d = [&](int c) { return c != tgt && (!mask_fn || sDot[c] < pos_dot); };
  auto counted_c = [&](int c) { return c != tgt && (!mask_fn || sCos[c] < cpos); };
  const int head = a.train_head;
  const bool all = a.all_heads != 0;
  const bool need_dot = all || head >= XFMR_LOSS_INFONCE;
  const bool need_cos = all || head <= XFMR_LOSS_CONTRASTIVE;
  TopK td{0u, 1.f, false}, tc{0u, 1.f, false};
  if (need_dot) td = select_topk(sDot, C, a.k_hard, counted_d, br);
  if (need_cos) tc = select_topk(sCos, C, a.k_hard, counted_c, br);

  // ---- 3. row reductions -------------------------------------------------------------------------------------
  const float sc2 = a.scale * kLog2e;
  const float chinge = pos_dot * (1.f - a.margin);
  float M = pos_dot * sc2;  // log-sum-exp shift: max over the positive and the counted negatives
  if (need_dot) {
    float mx = M;
    for (int c = tid; c < C; c += 256)
      if (counted_d(c) && td.weight(sDot[c]) > 0.f) mx = fmaxf(mx, sDot[c] * sc2);
    M = br.max(mx);
  }
  float cnt_d = 0.f, l = 0.f, nce = 0.f, hinge = 0.f, logi = 0.f, cnt_c = 0.f, contr = 0.f, ssum = 0.f, ssq = 0.f,
        smin = INFINITY, smax = -INFINITY, sw = 0.f;
  for (int c = tid; c < C; c += 256) {
    float w = 0.f;
    if (need_dot) {
      const float sv = sDot[c];
      const float md = counted_d(c) ? td.weight(sv) : 0.f;
      cnt_d += md;
      const float e = md > 0.f ? exp2f(sv * sc2 - M) * md : 0.f;
      l += e;
      if (head == XFMR_LOSS_INFONCE) w = e;
      nce = fmaf(xf_softplus(sv), md, nce);
      if (head == XFMR_LOSS_NCE) w = md * xf_sigmoid(sv);
      const float d = sv - chinge;
      hinge = fmaf(fmaxf(d, 0.f), md, hinge);
      if (head == XFMR_LOSS_PAIRWISE_HINGE) w = d > 0.f ? md : 0.f;
      logi = fmaf(xf_softplus(d), md, logi);
      if (head == XFMR_LOSS_PAIRWISE_LOGISTIC) w = md * xf_sigmoid(d);
      ssum = fmaf(sv, md, ssum);
      ssq = fmaf(sv * sv, md, ssq);
      if (md > 0.f) { smin = fminf(smin, sv); smax = fmaxf(smax, sv); }
    }
    if (need_cos) {
      const float cv = sCos[c];
      const float mc = counted_c(c) ? tc.weight(cv) : 0.f;
      cnt_c += mc;
      const float d = cv - 1.f + a.margin;
      contr = fmaf(fmaxf(d, 0.f), mc, contr);
      if (head == XFMR_LOSS_CONTRASTIVE || head == XFMR_LOSS_ALIGNMENT_CONTRASTIVE) w = d > 0.f ? mc * sRc[c] : 0.f;
    }
    sw += w;
    sW[c] = w;
  }
  cnt_d = br.sum(cnt_d); l = br.sum(l); nce = br.sum(nce); hinge = br.sum(hinge); logi = br.sum(logi);
  cnt_c = br.sum(cnt_c); contr = br.sum(contr); ssum = br.sum(ssum); ssq = br.sum(ssq); sw = br.sum(sw);
  smin = br.min(smin); smax = br.max(smax);  // (the last reduction also publishes sW)

  const float epos = exp2f(pos_dot * sc2 - M);
  const float ltot = l + epos;
  const float inv_d = 1.f / (cnt_d + 1e-9f), inv_c = 1.f / (cnt_c + 1e-9f);
  if (tid == 0) {
    double acc[BP];
#pragma unroll
    for (int k = 0; k < BP; ++k) acc[k] = 0.0;
    const float loss_align = 1.f - cpos, loss_contr = contr * inv_c;
    acc[XFMR_LOSS_ALIGNMENT] = loss_align;
    acc[XFMR_LOSS_ALIGNMENT_CONTRASTIVE] = loss_align + loss_contr;
    acc[XFMR_LOSS_CONTRASTIVE] = loss_contr;
    acc[XFMR_LOSS_INFONCE] = (M + log2f(ltot)) * kLn2 - a.scale * pos_dot;
    acc[XFMR_LOSS_NCE] = xf_softplus(-pos_dot) + nce * inv_d;
    acc[XFMR_LOSS_PAIRWISE_HINGE] = hinge * inv_d;
    acc[XFMR_LOSS_PAIRWISE_LOGISTIC] = logi * inv_d;
    int num_neg = C - 1;  // losses.py:387-390
    if (a.k_hard > 0 && a.k_hard < num_neg) num_neg = a.k_hard;
    acc[8] = (double)cnt_d / ((double)num_neg + 1e-9);
    acc[9] = pos_dot; acc[10] = (double)pos_dot * pos_dot;
    acc[11] = ssum; acc[12] = ssq; acc[13] = cnt_d; acc[14] = 1.0;
    acc[20] = pos_dot; acc[21] = pos_dot; acc[22] = smin; acc[23] = smax;
    if (!all) {  // only the train head was evaluated: the others are reported as 0
      for (int k = 0; k < XFMR_NUM_LOSSES; ++k)
        if (k != head) acc[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < BP; ++k) a.blockpart[(int64_t)k * gridDim.x + blockIdx.x] = acc[k];
  }

  // ---- 4a. gradient of the train head w.r.t. the candidates --------------------------------------------------
  // dot heads: dL/de_c = a_c q with a_c = dL/ds_c; cosine heads (logits on e_c / max(|e_c|, 1e-8)):
  // dL/de_c = b_c / |e_c| (q_hat - cos_c e_hat_c), b_c = dL/dcos_c. The masks and the top-k selection are piecewise
  // constant: no gradient flows through them (the reference's boolean masks / topk indices carry none either).
  if (a.d_cand) {
    const bool cosh_c = head <= XFMR_LOSS_CONTRASTIVE;
    float* dc = a.d_cand + row * (int64_t)C * H;
    const bool qclamped = sqrtf(qq) < 1e-8f;
    for (int64_t idx = tid; idx < (int64_t)C * H; idx += 256) {
      const int c = (int)(idx / H), h = (int)(idx - (int64_t)c * H);
      float coef;  // dL/d(logit of candidate c), in the head's own logit space
      const bool is_t = c == tgt;
      switch (head) {
        case XFMR_LOSS_INFONCE: coef = is_t ? -a.scale * (1.f - epos / ltot) : a.scale * sW[c] / ltot; break;
        case XFMR_LOSS_NCE: coef = is_t ? -xf_sigmoid(-pos_dot) : sW[c] * inv_d; break;
        case XFMR_LOSS_PAIRWISE_HINGE:
        case XFMR_LOSS_PAIRWISE_LOGISTIC: coef = is_t ? -(1.f - a.margin) * sw * inv_d : sW[c] * inv_d; break;
        case XFMR_LOSS_ALIGNMENT: coef = is_t ? -1.f : 0.f; break;
        // sW of the cosine heads carries the candidate's 1/|e| (for the query gradient): take it out again
        case XFMR_LOSS_CONTRASTIVE: coef = is_t ? 0.f : sW[c] * inv_c * fmaxf(1.f / sRc[c], 1e-8f); break;
        default: coef = is_t ? -1.f : sW[c] * inv_c * fmaxf(1.f / sRc[c], 1e-8f); break;  // ALIGNMENT_CONTRASTIVE
      }
      float g;
      if (!cosh_c) {
        g = coef * sQ[h];
      } else {
        const float rc = sRc[c], e = cand[idx];
        const float qh = sQ[h] * rq;  // q_hat (q / max(|q|, 1e-8))
        const bool cclamped = rc >= 1e8f;  // |e_c| < 1e-8: the norm was clamped, e_hat = e / 1e-8 is linear in e
        g = cclamped ? coef * rc * qh : coef * rc * (qh - (e * rc) * sCos[c]);
        (void)qclamped;
      }
      dc[idx] = g;
    }
  }

  // ---- 4. gradient of the train head w.r.t. the query ---------------------------------------------------------
  if (!a.d_query) return;
  const bool cosh = head <= XFMR_LOSS_CONTRASTIVE;
  constexpr int HPT = DENSE_MAX_H / 256;  // columns per thread: h = tid + 256 u
  float g[HPT];
  float dot_part = 0.f;
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int h = tid + 256 * u;
    g[u] = 0.f;
    if (h >= H) continue;
    float O = 0.f;
    if (head != XFMR_LOSS_ALIGNMENT)
      for (int c = 0; c < C; ++c) O = fmaf(sW[c], cand[(int64_t)c * H + h], O);
    const float e = cand[(int64_t)tgt * H + h];
    switch (head) {
      case XFMR_LOSS_INFONCE: g[u] = a.scale * (O / ltot - (1.f - epos / ltot) * e); break;
      case XFMR_LOSS_NCE: g[u] = -xf_sigmoid(-pos_dot) * e + O * inv_d; break;
      case XFMR_LOSS_PAIRWISE_HINGE:
      case XFMR_LOSS_PAIRWISE_LOGISTIC: g[u] = (O - (1.f - a.margin) * sw * e) * inv_d; break;
      case XFMR_LOSS_ALIGNMENT: g[u] = -rcpos * e; break;
      case XFMR_LOSS_CONTRASTIVE: g[u] = O * inv_c; break;
      default: g[u] = -rcpos * e + O * inv_c; break;  // ALIGNMENT_CONTRASTIVE
    }
    dot_part += g[u] * sQ[h] * rq;
  }
  if (cosh) {  // g is dL/dq_hat: apply the Jacobian of q / max(|q|, eps)
    const float dotp = br.sum(dot_part);
    const bool clamped = sqrtf(qq) < 1e-8f;
#pragma unroll
    for (int u = 0; u < HPT; ++u) {
      const int h = tid + 256 * u;
      if (h < H) g[u] = clamped ? g[u] * rq : rq * (g[u] - sQ[h] * rq * dotp);
    }
  }
#pragma unroll
  for (int u = 0; u < HPT; ++u) {
    const int h = tid + 256 * u;
    if (h < H) a.d_query[row * H + h] = g[u];
  }
}

__global__ void dense_counts_kernel(int* counts, int c, int n) {
  counts[0] = c;
  counts[1] = n;
  counts[2] = c;
}

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t xfmr_dense_loss_workspace(int64_t N, int32_t C, int32_t H) {
  if (N <= 0 || C <= 0 || H <= 0) return 0;
  return 256 + up256((size_t)N * BP * sizeof(double)) + 256;
}

int xfmr_dense_loss(const xfmr_loss_cfg* cfg, const float* query, const float* cand, const int64_t* target,
                    int32_t target_mode, int64_t N, int32_t C, int32_t H, float* losses, float* stats, float* d_query,
                    void* workspace, size_t workspace_bytes, void* stream) {
  return xfmr_dense_loss_grads(cfg, query, cand, target, target_mode, N, C, H, losses, stats, d_query, nullptr, workspace,
                               workspace_bytes, stream);
}

int xfmr_dense_loss_grads(const xfmr_loss_cfg* cfg, const float* query, const float* cand, const int64_t* target,
                          int32_t target_mode, int64_t N, int32_t C, int32_t H, float* losses, float* stats,
                          float* d_query, float* d_cand, void* workspace, size_t workspace_bytes, void* stream) {
  if (!cfg || !query || !cand || !losses || !stats || !workspace || N <= 0 || C <= 0 || H <= 0) return XFMR_EINVAL;
  if (target_mode < XFMR_TARGET_FIRST || target_mode > XFMR_TARGET_EXPLICIT) return XFMR_EINVAL;
  if ((target_mode == XFMR_TARGET_EXPLICIT) != (target != nullptr)) return XFMR_EINVAL;  // losses.py:233-238
  if (target_mode == XFMR_TARGET_DIAGONAL && N > C) return XFMR_EINVAL;
  if (cfg->train_head < 0 || cfg->train_head >= XFMR_NUM_LOSSES || cfg->num_hard_negatives < 0) return XFMR_EINVAL;
  if ((H & 3) || H > DENSE_MAX_H || C > DENSE_MAX_C) return XFMR_EUNSUPPORTED;
  if (!xf_aligned16(query) || !xf_aligned16(cand) || !xf_aligned16(workspace)) return XFMR_EALIGN;
  if (workspace_bytes < xfmr_dense_loss_workspace(N, C, H)) return XFMR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  unsigned char* ws = (unsigned char*)workspace;
  int* counts = (int*)ws;
  double* blockpart = (double*)(ws + 256);
  double* tot = (double*)(ws + 256 + up256((size_t)N * BP * sizeof(double)));
  DenseArgs a{};
  a.q = query; a.cand = cand; a.target = target; a.target_mode = target_mode; a.N = N; a.C = C; a.H = H;
  a.train_head = cfg->train_head; a.all_heads = cfg->all_heads; a.mask_fn = cfg->mask_false_negatives;
  a.k_hard = cfg->num_hard_negatives; a.scale = cfg->scale; a.margin = cfg->margin;
  a.d_query = d_query; a.blockpart = blockpart; a.d_cand = d_cand;
  const size_t smem = ((size_t)H + 4 * (size_t)C + 8) * sizeof(float);
  if (hipFuncSetAttribute((const void*)dense_loss_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL(dense_counts_kernel, dim3(1), dim3(1), 0, st, counts, (int)C, (int)N);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL(dense_loss_kernel, dim3((unsigned)N), dim3(256), smem, st, a);
  XF_LAUNCH_CHECK();
  return xf_loss_finalize(blockpart, (int)N, 1, counts, XFMR_NEG_SHARED, 0, (int64_t)C, losses, stats, tot, st);
}

}  // extern "C"
