// Shared between loss.hip (generic / fp32 main kernel, compaction, combine, host code) and the per-H
// translation units of the LDS-DMA bf16 main kernel (loss_dma_h*.hip).
#pragma once
#include "common.h"

namespace xfl {  // named: LossArgs crosses translation units

constexpr int BN = 64;        // negatives per tile
constexpr int QB = 128;       // queries per workgroup
constexpr int REC = 16;       // floats per (split, query) partial record
constexpr int LDT = BN + 4;   // transposed image [h][j] leading dimension (see attention.hip note)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
enum { R_CNTD = 0, R_M, R_L, R_NCE, R_HINGE, R_LOGI, R_CNTC, R_CONTR, R_SSUM, R_SSQ, R_SMIN, R_SMAX, R_SW,
       R_POSDOT, R_RQ, R_QQ };
constexpr int BP = 24;        // doubles per block-partial record
// kernel-template head code of the InfoNCE gradient pass with false-negative masking (the lean epilogue below)
constexpr int HEAD_INFONCE_MASKED = XFMR_NUM_LOSSES;
// InfoNCE WITHOUT false-negative masking (full-catalogue softmax) when the logging pass of the same call has already
// run: the row maximum of the counted logits is in its records (R_SMAX), so the running maximum is pinned at
// max(scale * pos, scale * smax) and the lean epilogue applies -- no online rescaling of the dQ accumulators.
constexpr int HEAD_INFONCE_PINNED = XFMR_NUM_LOSSES + 1;
// PairwiseLogisticLoss (BPR, BASELINE config 3) with false-negative masking over in-batch negatives, stripped like the
// InfoNCE case above (loss_epilogue_bpr_masked): the general epilogue spends ~24 issue slots per logit on it (it also
// feeds the hinge sum, resolves ties by substitution, tests validity), this one ~16.
constexpr int HEAD_BPR_MASKED = XFMR_NUM_LOSSES + 2;
// AlignmentContrastiveLoss (CCL, BASELINE config 5) / ContrastiveLoss with masking over in-batch negatives: the cosine
// heads in the query-norm-free form of the masked logging epilogue (loss_epilogue_cos_masked), ~9 issue slots per logit.
constexpr int HEAD_CCL_MASKED = XFMR_NUM_LOSSES + 3, HEAD_CONTR_MASKED = XFMR_NUM_LOSSES + 4;

struct LossArgs {
  const float* tok; const float* table; const float* rnorm; int64_t n_rows;
  const int* counts;      // [0] = N valid positions (negative columns of the reference), [1] = Np queries,
                          // [2] = Nd DISTINCT negative items (shared mode): the columns the kernels walk
  // Shared mode: in-batch negatives repeat items (N positions draw from V items; N = 25 600 vs V = 3 883 on
  // MovieLens-1M), and every per-column quantity -- logit, mask, each head's term, the gradient weight -- is a function
  // of the ITEM. The columns are therefore the distinct items (ascending id) with their multiplicity as weight:
  // sum_j f(s_ij) = sum_u mult_u f(s_iu), dQ_i = sum_u mult_u w_iu e_u. Exact, and Nd <= min(N, V) columns.
  const int* neg_item;    // [Nd] (shared mode) or null (catalogue mode: item j)
  const float* neg_rc;    // [Nd] inverse norm of each negative's row (shared mode)
  const float* neg_mult;  // [Nd] multiplicity of the item among the N negatives (shared mode; null: 1)
  const int* qrow; const int* qpos;
  float* part; float* partO;
  // num_hard_negatives (losses.py:295-330), generic kernel only: `dump` != null turns the launch into a logits dump
  // (dump[qi * dump_ld + j] = S^T tile values, qinfo[qi] = {pos_dot, 1/|q|}); `tau` != null restricts the negatives of
  // row qi to those at / above its thresholds tau[qi] = {tau_dot, rho_dot, tau_cos, rho_cos}
  float* dump; int64_t dump_ld; float2* qinfo; const float4* tau;
  const float* pin_part;  // HEAD_INFONCE_PINNED: the logging pass's split records (row maxima)
  int pin_nsplit;         // ... and how many splits that pass ran with
  // Gradient pass with ONE column split (this workgroup sees every column of its 128 queries): finish the rows here --
  // dL/dquery written straight to d_tok rows (qrow), no (split, query, H) partials, no gradient work left for the
  // combine kernel. Null: write the partials to partO as before.
  float* d_tok;
  int T; int nsplit;
  int H;                  // generic kernel: the rows' real width (= their stride); <= the kernel's template width, % 32 == 0
  int train_head, mask_fn, mode, need_grad;
  float scale, margin;
};

// ---- per-sub-block epilogue shared by both main kernels ----------------------------------------------------
// s: the 32x32 tile S^T (negative in the registers, query on the lane) on entry, the train head's gradient
// weights on exit. nid_sb / rc_sb: LDS side data (item id, inverse norm) of the sub-block's 32 negatives.
struct RowState {
  float cnt_d, m, l, nce, hinge, logi, cnt_c, contr, ssum, ssq, smin, smax, sw;
};
struct RowConst {
  float pos_dot, cpos, chinge, sc2, rq, margin;
  int pos_item, head;
  bool mask_fn, catalog, cos_head;
  bool hard;                             // top-k restriction active (per-row thresholds below)
  float tau_d, rho_d, tau_c, rho_c;
};
// HC: the train head as a compile-time constant (its gradient weight is produced), or -1 = values only.
// HARD: weight every counted negative by its top-k weight (1 above the row's threshold, rho at it, 0 below).
template <bool ALL, int HC, int NO, bool HARD = false>
__device__ __forceinline__ void loss_epilogue_t(f32x16& s, RowState& st, f32x16 (&o)[NO], const RowConst& k,
                                                const int* nid_sb, const float* rc_sb, const float* mu_sb, int hh) {
  float& cnt_d = st.cnt_d; float& m = st.m; float& l = st.l; float& nce = st.nce; float& hinge = st.hinge;
  float& logi = st.logi; float& cnt_c = st.cnt_c; float& contr = st.contr; float& ssum = st.ssum;
  float& ssq = st.ssq; float& smin = st.smin; float& smax = st.smax; float& sw = st.sw;
  const float pos_dot = k.pos_dot, cpos = k.cpos, chinge = k.chinge, sc2 = k.sc2, rq = k.rq;
  const int pos_item = k.pos_item;
  constexpr int head = HC;
  const bool mask_fn = k.mask_fn, catalog = k.catalog;
  constexpr bool cos_head = HC >= 0 && HC <= XFMR_LOSS_CONTRASTIVE;
  constexpr int H = NO * 32;  // only used to walk the gradient accumulators
  // The lane's 16 accumulator rows (r&3) + 8*(r>>2) + 4*hh are four runs of 4 consecutive negatives:
  // per-negative side data (item id, inverse norm) comes as one 16-byte LDS read per run and array.
  // HC == -2: the logging pass of a step whose InfoNCE value already comes from the gradient pass (all_heads = 2)
  const bool want_lse = (ALL && HC != -2) || head == XFMR_LOSS_INFONCE;
  if (want_lse && !mask_fn) {
    // online log-sum-exp: without false-negative masking a counted logit may exceed the running max
    // (with masking every counted logit is < the positive's, and m = scale * pos stays fixed)
    float bmax = m;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[8 * g + 4 * hh]);
      const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool same = nn[u] == pos_item;
        const float svp = same ? pos_dot : s[4 * g + u];
        const bool md = (nn[u] >= 0) & !(catalog & same) & (HARD ? svp >= k.tau_d : true);
        bmax = fmaxf(bmax, md ? svp * sc2 : m);
      }
    }
    bmax = fmaxf(bmax, xf_half_swap(bmax));
    if (__any(bmax > m)) {
      const float alpha = xf_exp2(m - bmax);
      l *= alpha;
      if (head == XFMR_LOSS_INFONCE) {
#pragma unroll
        for (int i = 0; i < H / 32; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        sw *= alpha;
      }
      m = bmax;
    }
  }
  // Every head's row reductions and the train head's gradient weight (left in s[r] for the second MFMA).
  // Branch-free per element; the `head` switches are wave-uniform.
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl0 = 8 * g + 4 * hh;
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[jl0]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    float rc[4] = {0.f, 0.f, 0.f, 0.f};
    if (ALL || cos_head) {
      const float4 c4 = *reinterpret_cast<const float4*>(&rc_sb[jl0]);
      rc[0] = c4.x; rc[1] = c4.y; rc[2] = c4.z; rc[3] = c4.w;
    }
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[jl0]);  // multiplicity of each column's item
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u;
      const bool valid = nn[u] >= 0;
      const bool same = nn[u] == pos_item;  // exact tie: the negative IS the positive item
      const float sv = same ? pos_dot : s[r];
      const bool excl = catalog & same;
      float md = (valid & (mask_fn ? (sv < pos_dot) : true) & !excl) ? mu[u] : 0.f;
      if (HARD) md *= sv > k.tau_d ? 1.f : (sv == k.tau_d ? k.rho_d : 0.f);
      float w = 0.f;
      cnt_d += md;
      if (want_lse) {
        // counted logits are <= m by construction; the clamp keeps an uncounted large logit from inf * 0
        const float e = xf_exp2(fminf(sv * sc2 - m, 0.f)) * md;
        l += e;
        if (head == XFMR_LOSS_INFONCE) w = e;
      }
      if (ALL || head == XFMR_LOSS_NCE) {
        const float t = xf_exp2(-fabsf(sv) * kLog2e);                    // exp(-|x|)
        // softplus(x) = max(x, 0) + log(1 + exp(-|x|))
        nce = fmaf(fmaxf(sv, 0.f) + kLn2 * xf_log2(1.f + t), md, nce);  // md = multiplicity (or a top-k fraction)
        if (head == XFMR_LOSS_NCE) w = md * xf_rcp(1.f + t) * (sv >= 0.f ? 1.f : t);  // sigmoid(x)
      }
      if (ALL || head == XFMR_LOSS_PAIRWISE_HINGE || head == XFMR_LOSS_PAIRWISE_LOGISTIC) {
        const float d = sv - chinge;
        hinge = fmaf(fmaxf(d, 0.f), md, hinge);
        if (head == XFMR_LOSS_PAIRWISE_HINGE) w = d > 0.f ? md : 0.f;
        if (ALL || head == XFMR_LOSS_PAIRWISE_LOGISTIC) {
          const float t = xf_exp2(-fabsf(d) * kLog2e);
          logi = fmaf(fmaxf(d, 0.f) + kLn2 * xf_log2(1.f + t), md, logi);
          if (head == XFMR_LOSS_PAIRWISE_LOGISTIC) w = md * xf_rcp(1.f + t) * (d >= 0.f ? 1.f : t);
        }
      }
      if (ALL || cos_head) {
        const float c = same ? cpos : sv * rq * rc[u];
        float mc = (valid & (mask_fn ? (c < cpos) : true) & !excl) ? mu[u] : 0.f;
        if (HARD) mc *= c > k.tau_c ? 1.f : (c == k.tau_c ? k.rho_c : 0.f);
        cnt_c += mc;
        const float d = c - 1.f + k.margin;
        contr = fmaf(fmaxf(d, 0.f), mc, contr);
        if (head == XFMR_LOSS_CONTRASTIVE || head == XFMR_LOSS_ALIGNMENT_CONTRASTIVE)
          w = d > 0.f ? mc * rc[u] : 0.f;
      }
      if (ALL) {
        ssum = fmaf(sv, md, ssum);
        ssq = fmaf(sv * sv, md, ssq);
        smin = fminf(smin, md > 0.f ? sv : INFINITY);
        smax = fmaxf(smax, md > 0.f ? sv : -INFINITY);
      }
      sw += w;
      s[r] = w;
    }
    // keep the scheduler from interleaving the four runs (it otherwise holds all 16 elements' temporaries
    // live at once and spills at 2 waves/SIMD)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The common case of the gradient pass, stripped to what it needs: InfoNCE head, false-negative masking on, every
// column of the sub-block a real negative. With masking the running maximum is pinned at the positive's logit
// (m = scale * pos), a counted logit is one with s < pos (an exact tie by item id is not), the gradient weight IS the
// softmax term (sum of weights == l) and the count is an integer add-with-carry: ~8 issue slots per element
// instead of ~23 (VALU issue, not the matrix pipe, bounds this kernel).
template <bool CHECK_VALID>
__device__ __forceinline__ void loss_epilogue_infonce_masked(f32x16& s, float& l, float& cnt, float pos_dot, float sc2,
                                                             float m, int pos_item, const int* nid_sb,
                                                             const float* mu_sb, int hh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[8 * g + 4 * hh]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[8 * g + 4 * hh]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u;
      bool counted = (s[r] < pos_dot) & (nn[u] != pos_item);
      if (CHECK_VALID) counted &= nn[u] >= 0;  // only the last tile of the range can hold past-the-end columns
      cnt += counted ? mu[u] : 0.f;
      const float e = xf_exp2(counted ? fmaf(s[r], sc2, -m) : -INFINITY) * mu[u];
      l += e;
      s[r] = e;
    }
  }
}

// PairwiseLogisticLoss, masking on, shared negatives (HEAD_BPR_MASKED): per counted logit d = s - (1 - margin) pos,
// term softplus(d) = ln 2 log2(1 + 2^y), y = d log2 e, weight sigmoid(d) = 2^y / (1 + 2^y) -- one exponential, one
// logarithm, one reciprocal; `logi2` holds the log2 sum (ln 2 applied once per row by the caller), a tie by item id is not
// counted (no logit substitution: with masking on the substituted value is never < pos), past-the-end columns are
// multiplicity-0 duplicates (no validity test: the staging loop of loss_dma.inc).
__device__ __forceinline__ void loss_epilogue_bpr_masked(f32x16& s, float& logi2, float& cnt, float& sw, float pos_dot,
                                                         float chinge, int pos_item, const int* nid_sb, const float* mu_sb,
                                                         int hh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[8 * g + 4 * hh]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[8 * g + 4 * hh]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u;
      const bool counted = (s[r] < pos_dot) & (nn[u] != pos_item);
      const float md = counted ? mu[u] : 0.f;
      cnt += md;
      const float e = xf_exp2(fminf((s[r] - chinge) * kLog2e, 100.f));
      const float one_e = 1.f + e;
      logi2 = fmaf(xf_log2(one_e), md, logi2);
      const float w = md * (e * xf_rcp(one_e));
      sw += w;
      s[r] = w;
    }
  }
}

// Cosine heads, masking on, shared negatives (HEAD_CCL_MASKED / HEAD_CONTR_MASKED). With sc = s / |e| (rc = 1 / |e|):
// counted <=> cos < cos_pos <=> sc < pos / |e_pos| (kpos_c); the hinge max(cos - 1 + margin, 0) = (1 / |q|) max(sc - kappa_c, 0)
// with kappa_c = (1 - margin) |q| -- `contr_q` holds the sum WITHOUT the 1 / |q| (applied once per row by the caller); the
// gradient weight of a column inside the margin is mult / |e| (dL / d q_hat = sum w e; the Jacobian of q_hat follows per
// row). A tie by item id is not counted; past-the-end columns are multiplicity-0 duplicates.
__device__ __forceinline__ void loss_epilogue_cos_masked(f32x16& s, float& contr_q, float& cnt_c, float kpos_c,
                                                         float kappa_c, int pos_item, const int* nid_sb,
                                                         const float* rc_sb, const float* mu_sb, int hh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl0 = 8 * g + 4 * hh;
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[jl0]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 c4 = *reinterpret_cast<const float4*>(&rc_sb[jl0]);
    const float rc[4] = {c4.x, c4.y, c4.z, c4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[jl0]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u;
      const float sc = s[r] * rc[u];
      const float mc = ((sc < kpos_c) & (nn[u] != pos_item)) ? mu[u] : 0.f;
      cnt_c += mc;
      const float dd = sc - kappa_c;
      contr_q = fmaf(fmaxf(dd, 0.f), mc, contr_q);
      s[r] = dd > 0.f ? mc * rc[u] : 0.f;
    }
  }
}

// The unmasked counterpart (HEAD_INFONCE_PINNED): every valid column counts (the positive's own item only in the
// in-batch form, with the positive's logit, as losses.py has it); m >= every counted logit by construction.
template <bool CHECK_VALID>
__device__ __forceinline__ void loss_epilogue_infonce_pinned(f32x16& s, float& l, float& cnt, float pos_dot, float sc2,
                                                             float m, int pos_item, bool catalog, const int* nid_sb,
                                                             const float* mu_sb, int hh) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[8 * g + 4 * hh]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[8 * g + 4 * hh]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = 4 * g + u;
      const bool same = nn[u] == pos_item;
      const float sv = same ? pos_dot : s[r];
      bool counted = !(catalog & same);
      if (CHECK_VALID) counted &= nn[u] >= 0;
      const float w = counted ? mu[u] : 0.f;
      cnt += w;
      const float e = xf_exp2(fminf(fmaf(sv, sc2, -m), 0.f)) * w;
      l += e;
      s[r] = e;
    }
  }
}

// The logging pass in its common case -- false-negative masking on, in-batch (shared) negatives, no top-k -- with the
// per-logit work cut from ~47 to ~36 issue slots (VALU issue bounds this pass: DESIGN.md section 4):
//   * an exact tie (the negative IS the positive item) is simply not counted: no logit substitution (with masking on the
//     substituted value s = pos is never < pos anyway);
//   * softplus(x) / ln 2 = log2(1 + 2^y), y = x log2(e) clamped at 100 (fp32 cannot tell 1 + 2^y from 2^y beyond y = 24,
//     and a masked-out logit must stay finite for the multiply by its zero weight): no max / |.| pair, the ln 2 is applied
//     once per row (st.nce, st.logi hold the log2 sums until loss_logging_masked_finish);
//   * cosine heads in the query-norm-free form: with sc = s / |e|, c < c_pos <=> sc < pos / |e_pos| and
//     max(c - 1 + margin, 0) = (1 / |q|) max(sc - (1 - margin) |q|, 0): one multiply per logit, 1 / |q| once per row;
//   * min / max through NaN (fminf / fmaxf = minNum / maxNum return the other operand): one select for both;
//   * ONE exponential per logit (SHARE): 2^(y - c) = 2^y 2^-c, so the pairwise-logistic term takes e = 2^y from the NCE
//     term and a per-row constant (as does the InfoNCE sum when the temperature is 1, the reference's default); the clamps
//     move from the exponent to the value (min(e k, 2^100), min(e k, 1)). Transcendentals issue at a quarter of the
//     vector rate: 5 -> 3 (4 -> 3 without the log-sum-exp) took 7-15 % off the pass. Rows whose constants leave
//     [2^-64, 2^64] (|logit of the positive| > 44) make their wave take the two-exponential form.
// HEAD code -3: without the InfoNCE log-sum-exp (its value comes from the gradient pass), -4: with it.
constexpr int HEAD_LOG_MASKED = -3, HEAD_LOG_MASKED_LSE = -4;
// -5: the same fast epilogue WITHOUT false-negative masking over the full catalogue (BASELINE config 4: every table row a
// column, mask_false_negatives = False, the positive's own column the only one excluded), without the log-sum-exp (the
// pinned gradient pass of the same call produces InfoNCE's value): MASK = false drops the two `< positive` compares.
constexpr int HEAD_LOG_UNMASKED_CATALOG = -5;
struct LogConst {
  float pos_dot, sc2, m, chinge, clog2e /* chinge * log2e */, kpos_c /* pos / |e_pos| */, kappa_c /* (1 - margin) |q| */;
  int pos_item;
  float kexp /* 2^-clog2e */, k2m /* 2^-m */;  // SHARE
};
template <bool CHECK_VALID, bool LSE, bool SHARE, bool MASK = true>
__device__ __forceinline__ void loss_epilogue_logging_masked(const f32x16& s, RowState& st, const LogConst& k,
                                                             const int* nid_sb, const float* rc_sb, const float* mu_sb,
                                                             int hh) {
  const float qnan = __builtin_nanf("");
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl0 = 8 * g + 4 * hh;
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[jl0]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 c4 = *reinterpret_cast<const float4*>(&rc_sb[jl0]);
    const float rc[4] = {c4.x, c4.y, c4.z, c4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[jl0]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float sv = s[4 * g + u];
      bool other = nn[u] != k.pos_item;              // not the positive's own item (past-the-end columns carry -1)
      if (CHECK_VALID) other &= nn[u] >= 0;
      const bool cd = MASK ? ((sv < k.pos_dot) & other) : other;
      const float md = cd ? mu[u] : 0.f;
      st.cnt_d += md;
      const float y = sv * kLog2e;
      const float e = xf_exp2(fminf(y, 100.f));
      if (LSE) {
        if (SHARE) st.l = fmaf(fminf(e * k.k2m, 1.f), md, st.l);  // (temperature 1: sc2 == log2 e)
        else st.l = fmaf(xf_exp2(fminf(fmaf(sv, k.sc2, -k.m), 0.f)), md, st.l);
      }
      st.nce = fmaf(xf_log2(1.f + e), md, st.nce);
      const float d = sv - k.chinge;
      st.hinge = fmaf(fmaxf(d, 0.f), md, st.hinge);
      if (SHARE) st.logi = fmaf(xf_log2(1.f + fminf(e * k.kexp, 0x1p100f)), md, st.logi);
      else st.logi = fmaf(xf_log2(1.f + xf_exp2(fminf(y - k.clog2e, 100.f))), md, st.logi);
      const float sc = sv * rc[u];
      const float mc = (MASK ? ((sc < k.kpos_c) & other) : other) ? mu[u] : 0.f;
      st.cnt_c += mc;
      st.contr = fmaf(fmaxf(sc - k.kappa_c, 0.f), mc, st.contr);
      const float t1 = sv * md;
      st.ssum += t1;
      st.ssq = fmaf(t1, sv, st.ssq);
      const float svm = cd ? sv : qnan;
      st.smin = fminf(st.smin, svm);
      st.smax = fmaxf(st.smax, svm);
    }
    __builtin_amdgcn_sched_barrier(0);  // (as in loss_epilogue_t: one run of four elements at a time)
  }
}
// The same epilogue on PAIRS of neighbouring logits (round 3). What the probe says of gfx950's vector pipe
// (scripts/probe/valu_rates.hip, profiles/r03_valu_rates.log): a plain fp32 instruction holds a SIMD 4 cycles per wave64 --
// not the guide's 2 --, a transcendental 8, and v_pk_{mul,add,fma}_f32 4.3-5.0 for TWO results. 16 of this epilogue's 30
// instructions per logit are mul / add / fma: written on float2 values they pack (-8 per logit); the sums become two
// partial sums each (even / odd columns), folded once per row. Scalars enter the packed operations as the LOW half of
// a broadcast (op_sel_hi), never the form scripts/check_isa.py rejects. In the SHARE form the argument of the one
// exponential is capped at min(100, clog2e + 100) instead of 100: then e * kexp <= 2^100 without its own clamp, and with
// masking on no counted logit can reach the cap (it is below the positive's, |m| <= 64: log_share).
#ifndef XFL_LOG_PK
#define XFL_LOG_PK 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct LogAcc {
  f32x2 cnt_d, l, nce, hinge, logi, cnt_c, contr, ssum, ssq;
};
__device__ __forceinline__ LogAcc log_acc_zero() {
  const f32x2 z = {0.f, 0.f};
  return LogAcc{z, z, z, z, z, z, z, z, z};
}
// (once per row. The odd half goes through an opaque move: left to itself the vectoriser writes x + y as one packed add
//  of the pair with ITSELF, halves swapped by op_sel -- the form check_isa.py rejects)
__device__ __forceinline__ float log_acc_sum2(f32x2 v) {
  float hi = v.y;
  asm volatile("" : "+v"(hi));
  return v.x + hi;
}
__device__ __forceinline__ void log_acc_fold(const LogAcc& a, RowState& st) {
  st.cnt_d += log_acc_sum2(a.cnt_d); st.l += log_acc_sum2(a.l); st.nce += log_acc_sum2(a.nce);
  st.hinge += log_acc_sum2(a.hinge); st.logi += log_acc_sum2(a.logi); st.cnt_c += log_acc_sum2(a.cnt_c);
  st.contr += log_acc_sum2(a.contr); st.ssum += log_acc_sum2(a.ssum); st.ssq += log_acc_sum2(a.ssq);
}
template <bool LSE, bool SHARE, bool MASK = true, bool PAIR_FENCE = false>
__device__ __forceinline__ void loss_epilogue_logging_masked_pk(const f32x16& s, RowState& st, LogAcc& la,
                                                                const LogConst& k, float ycap, const int* nid_sb,
                                                                const float* rc_sb, const float* mu_sb, int hh) {
  const float qnan = __builtin_nanf("");
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int jl0 = 8 * g + 4 * hh;
    const int4 n4 = *reinterpret_cast<const int4*>(&nid_sb[jl0]);
    const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
    const float4 c4 = *reinterpret_cast<const float4*>(&rc_sb[jl0]);
    const float rc[4] = {c4.x, c4.y, c4.z, c4.w};
    const float4 m4 = *reinterpret_cast<const float4*>(&mu_sb[jl0]);
    const float mu[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int u0 = 2 * p, u1 = 2 * p + 1;
      const f32x2 sv = {s[4 * g + u0], s[4 * g + u1]};
      const bool o0 = nn[u0] != k.pos_item, o1 = nn[u1] != k.pos_item;  // not the positive's own item
      const bool cd0 = MASK ? ((sv.x < k.pos_dot) & o0) : o0, cd1 = MASK ? ((sv.y < k.pos_dot) & o1) : o1;
      const f32x2 md = {cd0 ? mu[u0] : 0.f, cd1 ? mu[u1] : 0.f};
      la.cnt_d += md;
      const f32x2 y = sv * kLog2e;
      const float cap = (SHARE && MASK) ? ycap : 100.f;
      const f32x2 e = {xf_exp2(fminf(y.x, cap)), xf_exp2(fminf(y.y, cap))};
      if (LSE) {
        if (SHARE) {  // (temperature 1: sc2 == log2 e)
          const f32x2 t = e * k.k2m;
          const f32x2 tc = {fminf(t.x, 1.f), fminf(t.y, 1.f)};
          la.l = tc * md + la.l;
        } else {
          const f32x2 z = sv * k.sc2 - k.m;
          const f32x2 t = {xf_exp2(fminf(z.x, 0.f)), xf_exp2(fminf(z.y, 0.f))};
          la.l = t * md + la.l;
        }
      }
      const f32x2 e1 = e + 1.f;
      const f32x2 ln = {xf_log2(e1.x), xf_log2(e1.y)};
      la.nce = ln * md + la.nce;
      const f32x2 d = sv - k.chinge;
      const f32x2 dr = {fmaxf(d.x, 0.f), fmaxf(d.y, 0.f)};
      la.hinge = dr * md + la.hinge;
      f32x2 w;
      if (SHARE) {
        const f32x2 t = e * k.kexp;
        if (MASK) w = t + 1.f;
        else w = f32x2{fminf(t.x, 0x1p100f), fminf(t.y, 0x1p100f)} + 1.f;
      } else {
        const f32x2 z = y - k.clog2e;
        w = f32x2{xf_exp2(fminf(z.x, 100.f)), xf_exp2(fminf(z.y, 100.f))} + 1.f;
      }
      const f32x2 lw = {xf_log2(w.x), xf_log2(w.y)};
      la.logi = lw * md + la.logi;
      const f32x2 rc2 = {rc[u0], rc[u1]};
      const f32x2 sc = sv * rc2;
      const bool cc0 = MASK ? ((sc.x < k.kpos_c) & o0) : o0, cc1 = MASK ? ((sc.y < k.kpos_c) & o1) : o1;
      const f32x2 mc = {cc0 ? mu[u0] : 0.f, cc1 ? mu[u1] : 0.f};
      la.cnt_c += mc;
      const f32x2 dc = sc - k.kappa_c;
      const f32x2 dcr = {fmaxf(dc.x, 0.f), fmaxf(dc.y, 0.f)};
      la.contr = dcr * mc + la.contr;
      const f32x2 t1 = sv * md;
      la.ssum += t1;
      la.ssq = t1 * sv + la.ssq;
      const float s0 = cd0 ? sv.x : qnan, s1 = cd1 ? sv.y : qnan;
      st.smin = fminf(st.smin, fminf(s0, s1));
      st.smax = fmaxf(st.smax, fmaxf(s0, s1));
      if (PAIR_FENCE) __builtin_amdgcn_sched_barrier(0);  // one pair's temporaries at a time (register-bound variants)
    }
    __builtin_amdgcn_sched_barrier(0);  // (as in loss_epilogue_t: one run of four elements at a time)
  }
}
// per row, once: back to natural-log sums and the 1 / |q| of the contrastive term
__device__ __forceinline__ void loss_logging_masked_finish(RowState& st, float rq) {
  st.nce *= kLn2;
  st.logi *= kLn2;
  st.contr *= rq;
}

// runtime head -> compile-time head (one wave-uniform switch per sub-block instead of ~6 per element)
template <bool ALL, bool GRAD, int NO, bool HARD = false>
__device__ __forceinline__ void loss_epilogue_h(f32x16& s, RowState& st, f32x16 (&o)[NO], const RowConst& k,
                                                const int* nid_sb, const float* rc_sb, const float* mu_sb, int hh) {
  if (!GRAD) {
    if (ALL) return loss_epilogue_t<true, -1, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    // values of ONE head: its accumulators are selected by the head, no weights needed -> reuse the
    // weighted variants (the weight computation is dead code without the second product)
  }
  switch (k.head) {
    case XFMR_LOSS_ALIGNMENT: return loss_epilogue_t<ALL, XFMR_LOSS_ALIGNMENT, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    case XFMR_LOSS_ALIGNMENT_CONTRASTIVE:
      return loss_epilogue_t<ALL, XFMR_LOSS_ALIGNMENT_CONTRASTIVE, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    case XFMR_LOSS_CONTRASTIVE:
      return loss_epilogue_t<ALL, XFMR_LOSS_CONTRASTIVE, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    case XFMR_LOSS_INFONCE: return loss_epilogue_t<ALL, XFMR_LOSS_INFONCE, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    case XFMR_LOSS_NCE: return loss_epilogue_t<ALL, XFMR_LOSS_NCE, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    case XFMR_LOSS_PAIRWISE_HINGE:
      return loss_epilogue_t<ALL, XFMR_LOSS_PAIRWISE_HINGE, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
    default: return loss_epilogue_t<ALL, XFMR_LOSS_PAIRWISE_LOGISTIC, NO, HARD>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
  }
}
template <bool ALL, bool GRAD, int NO>
__device__ __forceinline__ void loss_epilogue(f32x16& s, RowState& st, f32x16 (&o)[NO], const RowConst& k,
                                              const int* nid_sb, const float* rc_sb, const float* mu_sb, int hh) {
  if (k.hard) loss_epilogue_h<ALL, GRAD, NO, true>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
  else loss_epilogue_h<ALL, GRAD, NO, false>(s, st, o, k, nid_sb, rc_sb, mu_sb, hh);
}

// halves of a lane pair (l, l^32) hold disjoint negatives of the same query: combine
__device__ __forceinline__ RowState merge_halves(RowState st) {
  st.cnt_d += xf_half_swap(st.cnt_d); st.l += xf_half_swap(st.l); st.nce += xf_half_swap(st.nce);
  st.hinge += xf_half_swap(st.hinge); st.logi += xf_half_swap(st.logi); st.cnt_c += xf_half_swap(st.cnt_c);
  st.contr += xf_half_swap(st.contr); st.ssum += xf_half_swap(st.ssum); st.ssq += xf_half_swap(st.ssq);
  st.sw += xf_half_swap(st.sw);
  st.smin = fminf(st.smin, xf_half_swap(st.smin)); st.smax = fmaxf(st.smax, xf_half_swap(st.smax));
  return st;
}
// ... then one lane writes the (split, query) record; st = the merged state
__device__ __forceinline__ void write_partial(const RowState& st, float* rec, bool writer, float pos_dot, float rq,
                                              float qq) {
  if (writer) {
    rec[R_CNTD] = st.cnt_d; rec[R_M] = st.m; rec[R_L] = st.l; rec[R_NCE] = st.nce; rec[R_HINGE] = st.hinge;
    rec[R_LOGI] = st.logi; rec[R_CNTC] = st.cnt_c; rec[R_CONTR] = st.contr; rec[R_SSUM] = st.ssum;
    rec[R_SSQ] = st.ssq; rec[R_SMIN] = st.smin; rec[R_SMAX] = st.smax; rec[R_SW] = st.sw;
    rec[R_POSDOT] = pos_dot; rec[R_RQ] = rq; rec[R_QQ] = qq;
  }
}

}  // namespace xfl
using namespace xfl;

// launchers of the LDS-DMA main kernel, one translation unit per hidden size (compile time)
// head: XFMR_LOSS_* = gradient pass of that head; -1 = logging pass (all heads + statistics, values only)
int xf_launch_loss_dma_64(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st);
int xf_launch_loss_dma_128(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st);
int xf_launch_loss_dma_256(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st);
int xf_loss_dma_256_hparts(const LossArgs& a, int head);  // dQ column parts of the H = 256 gradient pass (loss_dma_h256.hip)
int xf_launch_loss_dma_384(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st);
