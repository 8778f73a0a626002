// Exact top-k retrieval + ranking metrics for the validation path.
//
// The reference validates one user at a time: encode(history) -> LanceDB IVF_HNSW_PQ search over the item embeddings
// with the history prefiltered out -> top_k ids -> seven torchmetrics retrieval metrics on a synthesised score vector
// (xfmr_rec/trainer.py:186-211, 266-325; index.py:214-255; metrics.py:17-79). Here a batch of users is scored EXACTLY
// against the whole table (the ANN's limit of full probing):
//   topk_kernel     one workgroup per query: scores for every item (16 lanes per item, 16-byte pieces), the query's
//                   history and the padding row masked out, the k-th largest score by a 4-pass radix select, ordered
//                   compaction (ties: lower item index first), bitonic sort of the k survivors, best first
//   metrics_kernel  one thread per query: nDCG, MAP, AUROC, precision, recall, hit rate, MRR at k from the ranked
//                   list and the user's target set, with torchmetrics' definitions for a strictly decreasing score
//                   vector over [recommendations | missing targets] (metrics.py:66-79)
// metric: cosine (reference default, index.py:47), dot, or l2; score = 1 - distance as index.py:248-251 appends it.
#include "common.h"

namespace {

constexpr int TOPK_MAX = 1024;
constexpr int TOPK_MAX_H = 1024;

struct TopkArgs {
  const float* q; const float* table; const float* rnorm; int64_t n_rows;
  const int64_t* excl; const int64_t* excl_off;  // CSR of excluded item indices per query (may be null)
  float* scores;                                 // [B][n_rows] scratch
  int64_t* out_idx; float* out_score;
  int H, k, metric;
};

__device__ __forceinline__ unsigned key_of(float f) {
  const unsigned b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}

__global__ __launch_bounds__(256) void topk_kernel(TopkArgs a) {
  __shared__ __attribute__((aligned(16))) float sQ[TOPK_MAX_H];
  __shared__ unsigned hist[256];
  __shared__ int sh[4], wcnt[4], weq[4];
  __shared__ float sScore[TOPK_MAX];
  __shared__ int sIdx[TOPK_MAX];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t qi = blockIdx.x;
  const int H = a.H;
  const int64_t V = a.n_rows;
  float* sc = a.scores + qi * V;
  float qq = 0.f;
  for (int h = tid; h < H; h += 256) {
    const float v = a.q[qi * H + h];
    sQ[h] = v;
    qq += v * v;
  }
  qq = xf_wave_sum(qq);
  if (lane == 0) sh[w] = __float_as_int(qq);
  __syncthreads();
  qq = __int_as_float(sh[0]) + __int_as_float(sh[1]) + __int_as_float(sh[2]) + __int_as_float(sh[3]);
  const float rq = 1.f / fmaxf(sqrtf(qq), 1e-8f);
  __syncthreads();
  // ---- scores ------------------------------------------------------------------------------------------------
  {
    const int g = tid >> 4, j = tid & 15;
    for (int64_t it = g; it < V; it += 16) {
      const float* e = a.table + it * H;
      float dot = 0.f, ee = 0.f;
      for (int h = 4 * j; h < H; h += 64) {
        const float4 x = *reinterpret_cast<const float4*>(e + h);
        const float4 y = *reinterpret_cast<const float4*>(sQ + h);
        dot = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, dot))));
        ee = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, ee))));
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        dot += __shfl_xor(dot, o, 64);
        ee += __shfl_xor(ee, o, 64);
      }
      if (j == 0) {
        float s;
        if (a.metric == XFMR_METRIC_COSINE) s = dot * rq * a.rnorm[it];  // 1 - (1 - cos)
        else if (a.metric == XFMR_METRIC_DOT) s = dot;                   // 1 - (1 - dot)
        else s = 1.f - (qq - 2.f * dot + ee);                            // 1 - |q - e|^2
        sc[it] = it == 0 ? -INFINITY : s;                                // row 0 is the padding item
      }
    }
  }
  __syncthreads();
  if (a.excl) {
    for (int64_t x = a.excl_off[qi] + tid; x < a.excl_off[qi + 1]; x += 256) {
      const int64_t it = a.excl[x];
      if (it >= 0 && it < V) sc[it] = -INFINITY;
    }
  }
  __syncthreads();
  // ---- k-th largest finite score: 4-pass radix select ------------------------------------------------------------
  if (tid == 0) sh[3] = 0;
  __syncthreads();
  {
    int n = 0;
    for (int64_t it = tid; it < V; it += 256) n += sc[it] > -INFINITY ? 1 : 0;
    if (n) atomicAdd(&sh[3], n);
  }
  __syncthreads();
  const int avail = sh[3];
  const int k = a.k < avail ? a.k : avail;
  unsigned T = 0;
  int need = 0;
  if (avail > a.k) {
    unsigned prefix = 0;
    need = a.k;
    for (int p = 3; p >= 0; --p) {
      hist[tid] = 0;
      __syncthreads();
      const int shift = 8 * p;
      for (int64_t it = tid; it < V; it += 256) {
        const float v = sc[it];
        if (!(v > -INFINITY)) continue;
        const unsigned key = key_of(v);
        if (p == 3 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int cum = 0, d = 255;
        for (; d > 0; --d) {
          if (cum + (int)hist[d] >= need) break;
          cum += (int)hist[d];
        }
        sh[0] = d; sh[1] = need - cum;
      }
      __syncthreads();
      prefix |= (unsigned)sh[0] << shift;
      need = sh[1];
      __syncthreads();
    }
    T = prefix;  // take every key > T and the first `need` (by item index) of the keys == T
  }
  // ---- ordered compaction of the survivors ------------------------------------------------------------------------
  if (tid == 0) { sh[0] = 0; sh[1] = 0; }
  __syncthreads();
  for (int64_t i0 = 0; i0 < V; i0 += 256) {
    const int64_t it = i0 + tid;
    bool gt = false, eq = false;
    float v = -INFINITY;
    if (it < V) {
      v = sc[it];
      if (v > -INFINITY) {
        if (avail <= a.k) gt = true;
        else {
          const unsigned key = key_of(v);
          gt = key > T;
          eq = key == T;
        }
      }
    }
    const unsigned long long beq = __ballot(eq);
    if (lane == 0) weq[w] = __popcll(beq);
    __syncthreads();
    int eq_before = sh[1];
    for (int x = 0; x < w; ++x) eq_before += weq[x];
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    eq_before += __popcll(beq & below);
    const bool take = gt || (eq && eq_before < need);
    const unsigned long long bt = __ballot(take);
    if (lane == 0) wcnt[w] = __popcll(bt);
    __syncthreads();
    int at = sh[0];
    for (int x = 0; x < w; ++x) at += wcnt[x];
    at += __popcll(bt & below);
    if (take && at < TOPK_MAX) { sScore[at] = v; sIdx[at] = (int)it; }
    __syncthreads();
    if (tid == 0) {
      sh[0] += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
      sh[1] += weq[0] + weq[1] + weq[2] + weq[3];
    }
    __syncthreads();
  }
  // ---- bitonic sort, best first (score descending, then item index ascending) -------------------------------------
  int np2 = 1;
  while (np2 < k) np2 <<= 1;
  for (int i = k + tid; i < np2; i += 256) { sScore[i] = -INFINITY; sIdx[i] = 0x7fffffff; }
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int j = i ^ stride;
        if (j > i) {
          const bool up = (i & size) == 0;  // "up" blocks hold the better elements first
          const float si = sScore[i], sj = sScore[j];
          const int ii = sIdx[i], ij = sIdx[j];
          const bool i_better = si > sj || (si == sj && ii < ij);
          if (i_better != up) { sScore[i] = sj; sScore[j] = si; sIdx[i] = ij; sIdx[j] = ii; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < a.k; i += 256) {
    a.out_idx[qi * a.k + i] = i < k ? (int64_t)sIdx[i] : -1;  // fewer than k candidates: padded like metrics.py:61-64
    a.out_score[qi * a.k + i] = i < k ? sScore[i] : -INFINITY;
  }
}

// rec (B,k) ranked item indices (-1 = padding), targets CSR; out (B,7): ndcg, map, auroc, precision, recall, hit, mrr;
// valid[b] = 0 when the user has no target (the reference returns {} for it, metrics.py:58-59)
__global__ void metrics_kernel(const int64_t* rec, const int64_t* tgt, const int64_t* tgt_off, int B, int k, int top_k,
                               float* out, uint8_t* valid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int64_t* t = tgt + tgt_off[b];
  const int nt_raw = (int)(tgt_off[b + 1] - tgt_off[b]);
  float* o = out + (int64_t)b * 7;
  for (int i = 0; i < 7; ++i) o[i] = 0.f;
  // distinct targets (target_ids = set(target_ids), metrics.py:66)
  int nt = 0;
  for (int i = 0; i < nt_raw; ++i) {
    bool dup = false;
    for (int j = 0; j < i; ++j) dup |= t[j] == t[i];
    nt += dup ? 0 : 1;
  }
  valid[b] = nt > 0;
  if (nt == 0) return;
  const int64_t* r = rec + (int64_t)b * k;
  // the list torchmetrics sees: max(len(rec), top_k) slots of recommendations (padding never matches a target), then
  // the missing targets. Every metric below only looks at the first top_k slots (+ the number of targets).
  const int K = top_k;
  float dcg = 0.f, ap_sum = 0.f, rr = 0.f;
  int hits = 0, pairs = 0, neg_seen = 0;
  for (int i = 0; i < K; ++i) {
    bool rel = false;
    if (i < k && r[i] >= 0) {
      for (int j = 0; j < nt_raw; ++j) rel |= t[j] == r[i];
    }
    if (rel) {
      ++hits;
      dcg += 1.f / log2f((float)i + 2.f);
      ap_sum += (float)hits / (float)(i + 1);
      if (rr == 0.f) rr = 1.f / (float)(i + 1);
    } else {
      ++neg_seen;
    }
    if (rel) pairs += 0;  // (positives above this point are counted when the negatives below them arrive)
    else pairs += hits;   // each earlier positive outranks this negative
  }
  float idcg = 0.f;
  for (int i = 0; i < (nt < K ? nt : K); ++i) idcg += 1.f / log2f((float)i + 2.f);
  o[0] = idcg > 0.f ? dcg / idcg : 0.f;
  o[1] = hits > 0 ? ap_sum / (float)hits : 0.f;
  o[2] = (hits > 0 && neg_seen > 0) ? (float)pairs / ((float)hits * (float)neg_seen) : 0.f;
  o[3] = (float)hits / (float)K;
  o[4] = (float)hits / (float)nt;
  o[5] = hits > 0 ? 1.f : 0.f;
  o[6] = rr;
}

}  // namespace

extern "C" {

size_t xfmr_topk_workspace(int64_t n_query, int64_t n_rows) {
  if (n_query <= 0 || n_rows <= 0) return 0;
  return (size_t)n_query * (size_t)n_rows * sizeof(float);
}

int xfmr_topk(const float* query, const float* table, const float* table_rnorm, int64_t n_rows, int64_t n_query,
              int32_t H, const int64_t* exclude, const int64_t* exclude_offsets, int32_t k, int32_t metric,
              int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes, void* stream) {
  if (!query || !table || !table_rnorm || !out_idx || !out_score || !workspace) return XFMR_EINVAL;
  if (n_rows <= 0 || n_query <= 0 || H <= 0 || k <= 0) return XFMR_EINVAL;
  if ((exclude == nullptr) != (exclude_offsets == nullptr)) return XFMR_EINVAL;
  if (metric < XFMR_METRIC_COSINE || metric > XFMR_METRIC_L2) return XFMR_EINVAL;
  if ((H & 3) || H > TOPK_MAX_H || k > TOPK_MAX || n_rows >= (1ll << 31)) return XFMR_EUNSUPPORTED;
  if (!xf_aligned16(table) || !xf_aligned16(workspace)) return XFMR_EALIGN;
  if (workspace_bytes < xfmr_topk_workspace(n_query, n_rows)) return XFMR_EWORKSPACE;
  TopkArgs a{};
  a.q = query; a.table = table; a.rnorm = table_rnorm; a.n_rows = n_rows; a.excl = exclude; a.excl_off = exclude_offsets;
  a.scores = (float*)workspace; a.out_idx = out_idx; a.out_score = out_score; a.H = H; a.k = k; a.metric = metric;
  hipLaunchKernelGGL(topk_kernel, dim3((unsigned)n_query), dim3(256), 0, (hipStream_t)stream, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_retrieval_metrics(const int64_t* rec_idx, const int64_t* targets, const int64_t* target_offsets, int32_t n_query,
                           int32_t k, int32_t top_k, float* out, uint8_t* valid, void* stream) {
  if (!rec_idx || !targets || !target_offsets || !out || !valid || n_query <= 0 || k <= 0 || top_k <= 0)
    return XFMR_EINVAL;
  hipLaunchKernelGGL(metrics_kernel, dim3((n_query + 63) / 64), dim3(64), 0, (hipStream_t)stream, rec_idx, targets,
                     target_offsets, n_query, k, top_k, out, valid);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

}  // extern "C"
