// Device-side sequence sampler: SeqDataset.__getitem__ + collate of the reference (xfmr_rec/data.py:669-805) for a
// whole batch in one launch, so that batches are born in HBM (above ~1e4 sequences/s the
// reference's per-row numpy sampling -- a Python loop over positions, a set difference over the catalogue per row --
// is the bottleneck of the training path).
//
// One workgroup per batch row. For the row's history h[0..n) with labels lab[0..n):
//   positions  data.py:669-688: all of 0..n-2 if there are <= max_seq_length of them, else max_seq_length of them
//              uniformly without replacement, sorted: every position gets a hashed key, the L smallest keys win
//              (bitwise search for the L-th smallest key, ties by position), order-preserving compaction
//   positives  data.py:690-722: for each sampled position p a uniform draw among the later positions (within
//              pos_lookahead if > 0) whose label is positive, via the prefix count of labels; 0 if there is none
//   negatives  data.py:724-747: as many items as sampled positions, uniform over the catalogue minus the row's
//              history, without replacement while that set is large enough (rejection sampling against two bitmaps)
//   collate    data.py:789-805: rows right-padded with 0 to `width`
// The random stream is a counter-based hash of (seed, row id, purpose, counter): reproducible, independent of the
// launch geometry -- and necessarily different from numpy's generator: parity with the reference is distributional
// (tests compare invariants and frequencies against the numpy restatement in oracle/sampler.py).
#include "common.h"

namespace {

constexpr int MAX_HIST = 8192;

struct SampleArgs {
  const int64_t* items; const uint8_t* labels; const int64_t* offsets; const int64_t* rows;
  int64_t* hist_out; int64_t* pos_out; int64_t* neg_out;
  uint32_t* bitmaps;  // [B][2][words]
  int B, width, max_len, lookahead, words;
  int64_t n_items;
  uint64_t seed;
};

__device__ __forceinline__ uint32_t rnd(uint64_t seed, int64_t row, uint32_t purpose, uint32_t ctr) {
  uint32_t x = xf_hash32((uint32_t)seed ^ (uint32_t)(seed >> 32) * 0x9e3779b9u ^ (uint32_t)row * 0x85ebca6bu);
  x = xf_hash32(x ^ (purpose * 0xc2b2ae35u) ^ ctr);
  return xf_hash32(x + 0x27d4eb2fu * ctr);
}
// uniform integer in [0, n) from 32 random bits (multiply-shift: bias < n / 2^32)
__device__ __forceinline__ uint32_t below(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

__device__ int block_sum_i(int v, int* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void seq_sample_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) int smem[];
  __shared__ int red[4];
  __shared__ int s_sel, s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t row = a.rows[b];
  const int64_t o0 = a.offsets[row];
  const int n = (int)(a.offsets[row + 1] - o0);
  const int64_t* h = a.items + o0;
  const uint8_t* lab = a.labels + o0;
  int* P = smem;               // [n + 1] prefix count of positive labels
  int* posidx = P + (n + 1);   // [n] positions with a positive label, ascending
  int* sel = posidx + n;       // [width] sampled positions
  int64_t* ho = a.hist_out + (int64_t)b * a.width;
  int64_t* po = a.pos_out + (int64_t)b * a.width;
  int64_t* no = a.neg_out + (int64_t)b * a.width;
  const int m = n - 1;  // candidate positions 0..m-1 (the last item only ever serves as a positive)
  const int L = a.max_len < a.width ? a.max_len : a.width;

  // ---- positions -------------------------------------------------------------------------------------------
  uint32_t T = 0xffffffffu;
  int need_eq = 0;
  if (m > L) {
    // smallest T with count(key <= T) >= L, bit by bit from the top
    uint32_t t = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = t | ((1u << bit) - 1u);  // try leaving this bit clear
      int c = 0;
      for (int i = tid; i < m; i += 256) c += rnd(a.seed, row, 1u, (uint32_t)i) <= cand ? 1 : 0;
      if (block_sum_i(c, red) < L) t |= 1u << bit;
    }
    T = t;
    int lt = 0;
    for (int i = tid; i < m; i += 256) lt += rnd(a.seed, row, 1u, (uint32_t)i) < T ? 1 : 0;
    need_eq = L - block_sum_i(lt, red);  // how many of the keys == T are taken (lowest positions first)
  }
  if (tid == 0) { s_sel = 0; s_base = 0; }
  __syncthreads();
  // order-preserving compaction of the selected positions, 256 at a time
  for (int i0 = 0; i0 < m; i0 += 256) {
    const int i = i0 + tid;
    bool lt = false, eq = false;
    if (i < m) {
      if (m <= L) lt = true;
      else {
        const uint32_t k = rnd(a.seed, row, 1u, (uint32_t)i);
        lt = k < T;
        eq = k == T;
      }
    }
    // rank among the equal keys (by position) decides which of them are taken
    const unsigned long long beq = __ballot(eq);
    __shared__ int weq[4], wsel[4];
    if (lane == 0) weq[w] = __popcll(beq);
    __syncthreads();
    int eq_before = s_base;
    for (int k = 0; k < w; ++k) eq_before += weq[k];
    const unsigned long long below_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    eq_before += __popcll(beq & below_mask);
    const bool take = lt || (eq && eq_before < need_eq);
    const unsigned long long bt = __ballot(take);
    if (lane == 0) wsel[w] = __popcll(bt);
    __syncthreads();
    int at = s_sel;
    for (int k = 0; k < w; ++k) at += wsel[k];
    at += __popcll(bt & below_mask);
    if (take && at < a.width) sel[at] = i;
    __syncthreads();
    if (tid == 0) {
      s_sel += wsel[0] + wsel[1] + wsel[2] + wsel[3];
      s_base += weq[0] + weq[1] + weq[2] + weq[3];
    }
    __syncthreads();
  }
  const int cnt = s_sel < a.width ? s_sel : a.width;  // = min(max(n - 1, 0), L)

  // ---- prefix count of positive labels + list of positive positions ---------------------------------------------
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    const bool p = i < n && lab[i] != 0;
    const unsigned long long bp = __ballot(p);
    __shared__ int wp[4];
    if (lane == 0) wp[w] = __popcll(bp);
    __syncthreads();
    int before = s_base;
    for (int k = 0; k < w; ++k) before += wp[k];
    const unsigned long long below_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    before += __popcll(bp & below_mask);
    if (i < n) {
      P[i] = before;
      if (p) posidx[before] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += wp[0] + wp[1] + wp[2] + wp[3];
    __syncthreads();
  }
  if (tid == 0) P[n] = s_base;
  __syncthreads();

  // ---- history + positives, right-padded --------------------------------------------------------------------------
  for (int k = tid; k < a.width; k += 256) {
    int64_t hv = 0, pv = 0;
    if (k < cnt) {
      const int p = sel[k];
      hv = h[p];
      const int start = p + 1;
      const int end = a.lookahead > 0 ? (start + a.lookahead < n ? start + a.lookahead : n) : n;
      const int c = P[end] - P[start];
      if (c > 0) pv = h[posidx[P[start] + (int)below(rnd(a.seed, row, 2u, (uint32_t)k), (uint32_t)c)]];
    }
    ho[k] = hv;
    po[k] = pv;
  }

  // ---- negatives ----------------------------------------------------------------------------------------------------
  uint32_t* inhist = a.bitmaps + (int64_t)b * 2 * a.words;
  uint32_t* chosen = inhist + a.words;
  for (int k = tid; k < 2 * a.words; k += 256) inhist[k] = 0u;
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const int64_t it = h[i];
    if (it >= 1 && it <= a.n_items) atomicOr(&inhist[it >> 5], 1u << (it & 31));
  }
  __syncthreads();
  int distinct = 0;
  for (int k = tid; k < a.words; k += 256) distinct += __popc(inhist[k]);
  distinct = block_sum_i(distinct, red);
  if (tid == 0) {
    const int64_t n_cand = a.n_items - distinct;
    const bool any_item = n_cand == 0;          // data.py:743-744: nothing left -> the whole catalogue
    const bool replace = !any_item ? n_cand < cnt : a.n_items < cnt;  // data.py:745-747
    uint32_t ctr = 0;
    for (int k = 0; k < cnt; ++k) {
      int64_t x = 0;
      for (int attempt = 0;; ++attempt) {
        x = 1 + (int64_t)below(rnd(a.seed, row, 3u, ctr++), (uint32_t)a.n_items);
        const uint32_t bit = 1u << (x & 31);
        const bool bad = (!any_item && (inhist[x >> 5] & bit)) || (!replace && (chosen[x >> 5] & bit));
        if (!bad) break;
        if (attempt >= 64) {  // crowded catalogue: walk forward from the draw to the first admissible item
          for (int64_t s = 0; s < a.n_items; ++s) {
            const int64_t y = 1 + (x - 1 + s) % a.n_items;
            const uint32_t by = 1u << (y & 31);
            if (!((!any_item && (inhist[y >> 5] & by)) || (!replace && (chosen[y >> 5] & by)))) { x = y; break; }
          }
          break;
        }
      }
      chosen[x >> 5] |= 1u << (x & 31);
      no[k] = x;
    }
    for (int k = cnt; k < a.width; ++k) no[k] = 0;
  }
}

size_t words_for(int64_t n_items) { return (size_t)((n_items + 1 + 31) / 32); }

}  // namespace

extern "C" {

size_t xfmr_seq_sample_workspace(int32_t batch, int64_t n_items) {
  if (batch <= 0 || n_items <= 0) return 0;
  return (size_t)batch * 2 * words_for(n_items) * sizeof(uint32_t);
}

int xfmr_seq_sample(const int64_t* items, const uint8_t* labels, const int64_t* offsets, const int64_t* rows,
                    int32_t batch, int32_t width, int32_t max_seq_length, int32_t pos_lookahead, int64_t n_items,
                    int32_t max_history, uint64_t seed, int64_t* hist_out, int64_t* pos_out, int64_t* neg_out,
                    void* workspace, size_t workspace_bytes, void* stream) {
  if (!items || !labels || !offsets || !rows || !hist_out || !pos_out || !neg_out || !workspace) return XFMR_EINVAL;
  if (batch <= 0 || width <= 0 || max_seq_length <= 0 || pos_lookahead < 0 || n_items <= 0 || max_history <= 0)
    return XFMR_EINVAL;
  if (n_items >= (1ll << 31) || max_history > MAX_HIST) return XFMR_EUNSUPPORTED;
  if (workspace_bytes < xfmr_seq_sample_workspace(batch, n_items)) return XFMR_EWORKSPACE;
  SampleArgs a{};
  a.items = items; a.labels = labels; a.offsets = offsets; a.rows = rows;
  a.hist_out = hist_out; a.pos_out = pos_out; a.neg_out = neg_out; a.bitmaps = (uint32_t*)workspace;
  a.B = batch; a.width = width; a.max_len = max_seq_length; a.lookahead = pos_lookahead;
  a.words = (int)words_for(n_items); a.n_items = n_items; a.seed = seed;
  const size_t smem = ((size_t)2 * max_history + 1 + width + 8) * sizeof(int);
  if (hipFuncSetAttribute((const void*)seq_sample_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return XFMR_EHIP;
  hipLaunchKernelGGL(seq_sample_kernel, dim3(batch), dim3(256), smem, (hipStream_t)stream, a);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

}  // extern "C"
