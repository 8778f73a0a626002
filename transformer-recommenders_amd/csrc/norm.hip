// HBM-bound row kernels: item-embedding gather + BertEmbeddings LayerNorm, LayerNorm fwd/bwd,
// embedding-parameter gradients, masked mean pooling, AdamW, per-item inverse norms.
// One wavefront owns one row (H <= 1024): lanes stride the row in 4-byte steps, so every wave
// instruction touches 256 contiguous bytes; row statistics are wave reductions, never LDS.
#include <stdlib.h>

#include "internal.h"

// single-level deterministic column sum dst[c] = sum_r src[r*cols + c] (gemm.hip)

namespace {

constexpr int kMaxPerLane = 16;  // H <= 1024

// ---- LayerNorm forward (optionally fused with the embedding gather) ---------------------------
struct LnFwdArgs {
  const float* x;          // [rows,H] input (when !GATHER)
  const int64_t* idx;      // GATHER: item index per row
  const float* table; int64_t n_rows;
  const float* pos_emb; const float* type_emb; int L;
  const int32_t* row_pos;  // GATHER, packed rows (xfmr_encoder_cfg.row_pos): the row's position; null: row % L
  const float* gamma; const float* beta;
  float* y; float* pre; float* mean; float* rstd; uint8_t* key_mask;
  int64_t rows; int H; float eps;
  XfDropout drop;
  __bf16* y16;             // optional bf16 copy of y: the A / B operand of the GEMMs that consume the output
};

template <int NPL, bool GATHER>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a_in) {
  LnFwdArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  float v[NPL];
  const int H = a.H;
  if (GATHER) {
    int64_t item = a.idx[row];
    if (item < 0 || item >= a.n_rows) item = 0;
    const float* src = a.table + item * H;
    const float* pe = a.pos_emb + (int64_t)(a.row_pos ? a.row_pos[row] : (int)(row % a.L)) * H;
    bool nz = false;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      float e = 0.f;
      if (c < H) {
        e = src[c];
        nz |= (e != 0.f);
        e = (e + a.type_emb[c]) + pe[c];  // same association as TF:modeling_bert.py:100-104
        a.pre[row * H + c] = e;
      }
      v[i] = e;
    }
    const bool any_nz = __any(nz);
    if (lane == 0) a.key_mask[row] = any_nz ? 1 : 0;
  } else {
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < H) ? a.x[row * H + c] : 0.f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) s += v[i];
  const float mean = xf_wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    const float d = (c < H) ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float var = xf_wave_sum(q) / (float)H;
  const float rstd = rsqrtf(var + a.eps);
  if (lane == 0) {
    a.mean[row] = mean;
    a.rstd[row] = rstd;
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    if (c < H) {
      float o = (v[i] - mean) * rstd * a.gamma[c] + a.beta[c];
      if (a.drop.on) o *= xf_keep_scale_2d(a.drop, (uint32_t)row, (uint32_t)c);
      a.y[row * H + c] = o;
      if (a.y16) a.y16[row * H + c] = (__bf16)o;
    }
  }
}

// ---- LayerNorm backward --------------------------------------------------------------------------
// dy is the gradient of the (possibly dropped-out, embedding site only) LayerNorm output.
struct LnBwdArgs {
  const float* dy; const float* x; const float* mean; const float* rstd; const float* gamma;
  float* dx; void* d_lin; float* partials;  // partials [blocks][3][H]; d_lin optional, bf16 if lin16
  int64_t rows; int H; int rows_per_block; int lin16;
  XfDropout drop_out;  // dropout that was applied to y itself (embedding site); off otherwise
  XfDropout drop_lin;  // dropout of the Linear output feeding this LayerNorm's input
};

template <int NPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a_in) {
  LnBwdArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop_out = xf_drop_resolve(a.drop_out); a.drop_lin = xf_drop_resolve(a.drop_lin);
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [4][3][H]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int H = a.H;
  float gam[NPL], dgam[NPL], dbet[NPL], dbias[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    gam[i] = (c < H) ? a.gamma[c] : 0.f;
    dgam[i] = dbet[i] = dbias[i] = 0.f;
  }
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  const int64_t r1 = (r0 + a.rows_per_block < a.rows) ? r0 + a.rows_per_block : a.rows;
  for (int64_t row = r0 + wid; row < r1; row += 4) {
    const float mean = a.mean[row], rstd = a.rstd[row];
    float xh[NPL], g[NPL];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      float dyv = 0.f, xv = 0.f;
      if (c < H) {
        dyv = a.dy[row * H + c];
        if (a.drop_out.on) dyv *= xf_keep_scale_2d(a.drop_out, (uint32_t)row, (uint32_t)c);
        xv = (a.x[row * H + c] - mean) * rstd;
      }
      xh[i] = xv;
      g[i] = dyv * gam[i];
      dgam[i] += dyv * xv;
      dbet[i] += dyv;
      sg += g[i];
      sgx += g[i] * xv;
    }
    const float mg = xf_wave_sum(sg) / (float)H;
    const float mgx = xf_wave_sum(sgx) / (float)H;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      if (c < H) {
        const float d = rstd * (g[i] - mg - xh[i] * mgx);
        a.dx[row * H + c] = d;
        float dl = d;
        if (a.drop_lin.on) dl = d * xf_keep_scale_2d(a.drop_lin, (uint32_t)row, (uint32_t)c);
        if (a.d_lin) {
          if (a.lin16) reinterpret_cast<__bf16*>(a.d_lin)[row * H + c] = (__bf16)dl;
          else reinterpret_cast<float*>(a.d_lin)[row * H + c] = dl;
        }
        dbias[i] += dl;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    if (c < H) {
      smem[(wid * 3 + 0) * H + c] = dgam[i];
      smem[(wid * 3 + 1) * H + c] = dbet[i];
      smem[(wid * 3 + 2) * H + c] = dbias[i];
    }
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 3 * H; o += 256) {
    const float s = smem[o] + smem[3 * H + o] + smem[6 * H + o] + smem[9 * H + o];
    a.partials[(int64_t)blockIdx.x * 3 * H + o] = s;
  }
}

// ---- vectorised variants for H = 4 * LPR (LPR = 16, 32, 64 lanes per row): 16 bytes per lane, several rows per
// wave instruction (a 1-KiB wave load instead of 256 B), row statistics by shuffles inside the LPR-lane group.
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, bool GATHER>
__global__ __launch_bounds__(256) void ln_fwd_v4_kernel(const LnFwdArgs a_in) {
  LnFwdArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop = xf_drop_resolve(a.drop);
  constexpr int H = 4 * LPR, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPR, li = lane % LPR, c = li * 4;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wid) * RPW + sub;
  const bool valid = row < a.rows;
  float4 v = make_float4(0, 0, 0, 0);
  if (GATHER) {
    int nz = 0;
    if (valid) {
      int64_t item = a.idx[row];
      if (item < 0 || item >= a.n_rows) item = 0;
      const float4 e = *reinterpret_cast<const float4*>(a.table + item * H + c);
      nz = (e.x != 0.f) | (e.y != 0.f) | (e.z != 0.f) | (e.w != 0.f);
      const float4 ty = *reinterpret_cast<const float4*>(a.type_emb + c);
      const float4 pe = *reinterpret_cast<const float4*>(a.pos_emb + (int64_t)(a.row_pos ? a.row_pos[row] : (int)(row % a.L)) * H + c);
      v.x = (e.x + ty.x) + pe.x; v.y = (e.y + ty.y) + pe.y; v.z = (e.z + ty.z) + pe.z; v.w = (e.w + ty.w) + pe.w;
      *reinterpret_cast<float4*>(a.pre + row * H + c) = v;
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) nz |= __shfl_xor(nz, o, 64);
    if (valid && li == 0) a.key_mask[row] = nz ? 1 : 0;
  } else if (valid) {
    v = *reinterpret_cast<const float4*>(a.x + row * H + c);
  }
  const float mean = row_sum<LPR>((v.x + v.y) + (v.z + v.w)) / (float)H;
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = row_sum<LPR>((dx * dx + dy * dy) + (dz * dz + dw * dw)) / (float)H;
  const float rstd = rsqrtf(var + a.eps);
  if (!valid) return;
  if (li == 0) {
    a.mean[row] = mean;
    a.rstd[row] = rstd;
  }
  const float4 g = *reinterpret_cast<const float4*>(a.gamma + c);
  const float4 b = *reinterpret_cast<const float4*>(a.beta + c);
  float4 o;
  o.x = dx * rstd * g.x + b.x; o.y = dy * rstd * g.y + b.y; o.z = dz * rstd * g.z + b.z; o.w = dw * rstd * g.w + b.w;
  if (a.drop.on) {
    xf_drop4(a.drop, (uint32_t)row, (uint32_t)(c), o);
  }
  *reinterpret_cast<float4*>(a.y + row * H + c) = o;
  if (a.y16) xf_st4<true>(a.y16, row * H + c, o);
}

template <int LPR>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const LnBwdArgs a_in) {
  LnBwdArgs a = a_in;  // (device-side step counter -> dropout key: xf_drop_resolve)
  XF_CHAIN_PRIO();
  a.drop_out = xf_drop_resolve(a.drop_out); a.drop_lin = xf_drop_resolve(a.drop_lin);
  constexpr int H = 4 * LPR, RPW = 64 / LPR;
  __shared__ float red[4][3][H];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int sub = lane / LPR, li = lane % LPR, c = li * 4;
  const float4 gam = *reinterpret_cast<const float4*>(a.gamma + c);
  float4 dgam = make_float4(0, 0, 0, 0), dbet = dgam, dbias = dgam;
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  const int64_t r1 = (r0 + a.rows_per_block < a.rows) ? r0 + a.rows_per_block : a.rows;
  for (int64_t rb = r0 + wid * RPW; rb < r1; rb += 4 * RPW) {  // wave-uniform trip count
    const int64_t row = rb + sub;
    const bool valid = row < r1;
    float4 dy = make_float4(0, 0, 0, 0), xv = dy;
    float mean = 0.f, rstd = 0.f;
    if (valid) {
      dy = *reinterpret_cast<const float4*>(a.dy + row * H + c);
      xv = *reinterpret_cast<const float4*>(a.x + row * H + c);
      mean = a.mean[row];
      rstd = a.rstd[row];
      if (a.drop_out.on) {
        xf_drop4(a.drop_out, (uint32_t)row, (uint32_t)(c), dy);
      }
    }
    float4 xh, g;
    xh.x = (xv.x - mean) * rstd; xh.y = (xv.y - mean) * rstd; xh.z = (xv.z - mean) * rstd; xh.w = (xv.w - mean) * rstd;
    g.x = dy.x * gam.x; g.y = dy.y * gam.y; g.z = dy.z * gam.z; g.w = dy.w * gam.w;
    dgam.x += dy.x * xh.x; dgam.y += dy.y * xh.y; dgam.z += dy.z * xh.z; dgam.w += dy.w * xh.w;
    dbet.x += dy.x; dbet.y += dy.y; dbet.z += dy.z; dbet.w += dy.w;
    const float mg = row_sum<LPR>((g.x + g.y) + (g.z + g.w)) / (float)H;
    const float mgx = row_sum<LPR>((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w)) / (float)H;
    if (valid) {
      float4 d;
      d.x = rstd * (g.x - mg - xh.x * mgx); d.y = rstd * (g.y - mg - xh.y * mgx);
      d.z = rstd * (g.z - mg - xh.z * mgx); d.w = rstd * (g.w - mg - xh.w * mgx);
      *reinterpret_cast<float4*>(a.dx + row * H + c) = d;
      float4 dl = d;
      if (a.drop_lin.on) {
        xf_drop4(a.drop_lin, (uint32_t)row, (uint32_t)(c), dl);
      }
      if (a.d_lin) {
        if (a.lin16) xf_st4<true>(a.d_lin, row * H + c, dl);
        else xf_st4<false>(a.d_lin, row * H + c, dl);
      }
      dbias.x += dl.x; dbias.y += dl.y; dbias.z += dl.z; dbias.w += dl.w;
    }
  }
  // combine the RPW row groups of the wave (lanes with equal li), then the 4 waves, in a fixed order
  auto fold = [&](float4& v) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      v.x += __shfl_xor(v.x, o, 64); v.y += __shfl_xor(v.y, o, 64);
      v.z += __shfl_xor(v.z, o, 64); v.w += __shfl_xor(v.w, o, 64);
    }
  };
  fold(dgam); fold(dbet); fold(dbias);
  if (sub == 0) {
    *reinterpret_cast<float4*>(&red[wid][0][c]) = dgam;
    *reinterpret_cast<float4*>(&red[wid][1][c]) = dbet;
    *reinterpret_cast<float4*>(&red[wid][2][c]) = dbias;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 3 * H; o += 256) {
    const float* r = &red[0][0][0];
    a.partials[(int64_t)blockIdx.x * 3 * H + o] = (r[o] + r[3 * H + o]) + (r[6 * H + o] + r[9 * H + o]);
  }
}

// sums `blocks` partial records [blocks][3][H] into up to three destinations: 64 outputs per workgroup,
// 16 record groups with 4 loads in flight each, fixed combine order (deterministic)
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* partials, int blocks, int H, float* d_gamma,
                                                             float* d_beta, float* d_bias) {
  __shared__ float red[16][64];
  const int c64 = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + c64;
  const int64_t stride = (int64_t)3 * H;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (o < 3 * H) {
    int b = rg;
    for (; b + 48 < blocks; b += 64) {
      s0 += partials[b * stride + o];
      s1 += partials[(b + 16) * stride + o];
      s2 += partials[(b + 32) * stride + o];
      s3 += partials[(b + 48) * stride + o];
    }
    for (; b < blocks; b += 16) s0 += partials[b * stride + o];
  }
  red[rg][c64] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && o < 3 * H) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][c64];
    const int which = o / H, c = o % H;
    float* dst = which == 0 ? d_gamma : which == 1 ? d_beta : d_bias;
    if (dst) dst[c] = s;
  }
}

// d_pos[t,c] = sum_b d_pre[b,t,c]  (t < L), 0 for L <= t < max_pos
__global__ void embed_pos_grad_kernel(const float* d_pre, float* d_pos, int B, int L, int H, int max_pos) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = blockIdx.y;
  if (c >= H) return;
  float s = 0.f;
  if (t < L)
    for (int b = 0; b < B; ++b) s += d_pre[((int64_t)b * L + t) * H + c];
  d_pos[(int64_t)t * H + c] = s;
}
__global__ void embed_type_grad_kernel(const float* d_pos, float* d_type, int L, int H) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  float s = 0.f;
  for (int t = 0; t < L; ++t) s += d_pos[(int64_t)t * H + c];
  d_type[c] = s;
  d_type[H + c] = 0.f;
}

// sentence-transformers Pooling over the token axis (modes the reference's ModelConfig.pooling_mode allows,
// models.py:47). One thread per (sequence, column).
//   mean:      sum_t tok*m / max(sum_t m, 1e-9)
//   max:       max_t (m ? tok : -1e9)
//   cls:       tok[0]
//   lasttoken: tok[last t with m != 0] (t = 0 when the row has none), times its mask value
__global__ void pool_kernel(const float* tok, const uint8_t* mask, float* out, int L, int H, int mode) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= H) return;
  const float* x = tok + (int64_t)b * L * H + c;
  const uint8_t* m = mask + (int64_t)b * L;
  float r;
  if (mode == XFMR_POOL_MEAN) {
    float s = 0.f, n = 0.f;
    for (int t = 0; t < L; ++t) {
      const float w = m[t] ? 1.f : 0.f;
      s += x[(int64_t)t * H] * w;
      n += w;
    }
    r = s / fmaxf(n, 1e-9f);
  } else if (mode == XFMR_POOL_MAX) {
    r = -INFINITY;
    for (int t = 0; t < L; ++t) r = fmaxf(r, m[t] ? x[(int64_t)t * H] : -1e9f);
  } else if (mode == XFMR_POOL_CLS) {
    r = x[0];
  } else {
    int last = -1;
    for (int t = 0; t < L; ++t)
      if (m[t]) last = t;
    r = last >= 0 ? x[(int64_t)last * H] : 0.f;
  }
  out[(int64_t)b * H + c] = r;
}

// y = x / max(|x|, eps) per row (torch.nn.functional.normalize, models.py:393-394; sentence-transformers
// Normalize, models.py:146-147). One wave per row.
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* x, float* y, float* inv_norm, int64_t rows, int H,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float v = x[row * H + c];
    s += v * v;
  }
  s = xf_wave_sum(s);
  const float inv = 1.f / fmaxf(sqrtf(s), eps);
  for (int c = lane; c < H; c += 64) y[row * H + c] = x[row * H + c] * inv;
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}
// dx = inv * (dy - y (y . dy)); with the norm clamped at eps the map is linear: dx = dy / eps
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* dy, const float* y, const float* inv_norm,
                                                         float* dx, int64_t rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float inv = inv_norm[row];
  const bool clamped = inv * eps >= 1.f;
  float d = 0.f;
  for (int c = lane; c < H; c += 64) d += y[row * H + c] * dy[row * H + c];
  d = clamped ? 0.f : xf_wave_sum(d);
  for (int c = lane; c < H; c += 64) dx[row * H + c] = inv * (dy[row * H + c] - y[row * H + c] * d);
}

// step_dev != nullptr: the step count comes from device memory (a captured step replays with the right bias
// corrections): step = *step_dev + step_off, corrections evaluated in double as the host does for the by-value form.
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2_sqrt, float gscale, const uint32_t* step_dev,
                             int step_off) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  if (step_dev) {
    const double t = (double)((int64_t)*step_dev + step_off);
    bc1 = (float)(1.0 - pow((double)b1, t));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, t));
  }
  if (i + 3 < n) {
    float4 pp = *reinterpret_cast<float4*>(p + i);
    const float4 gg = *reinterpret_cast<const float4*>(g + i);
    float4 mm = *reinterpret_cast<float4*>(m + i);
    float4 vv = *reinterpret_cast<float4*>(v + i);
    float* pa = &pp.x; const float* ga = &gg.x; float* ma = &mm.x; float* va = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gr = ga[j] * gscale;
      float pj = pa[j] * (1.f - lr * wd);
      ma[j] = b1 * ma[j] + (1.f - b1) * gr;
      va[j] = b2 * va[j] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(va[j]) / bc2_sqrt + eps;
      pa[j] = pj - (lr / bc1) * (ma[j] / denom);
    }
    *reinterpret_cast<float4*>(p + i) = pp;
    *reinterpret_cast<float4*>(m + i) = mm;
    *reinterpret_cast<float4*>(v + i) = vv;
  } else {
    for (int64_t k = i; k < n; ++k) {
      const float gr = g[k] * gscale;
      float pj = p[k] * (1.f - lr * wd);
      m[k] = b1 * m[k] + (1.f - b1) * gr;
      v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
      const float denom = sqrtf(v[k]) / bc2_sqrt + eps;
      p[k] = pj - (lr / bc1) * (m[k] / denom);
    }
  }
}

// x *= *s (the upstream gradient of a loss, a device scalar: no host sync). `loss.backward()` passes exactly 1: the
// workgroup then leaves without touching x (22 us and 105 MB per step at the benchmark shape otherwise).
__global__ __launch_bounds__(256) void scale_kernel(float* x, int64_t n4, int64_t n, const float* s) {
  const float f = s[0];
  if (f == 1.f) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) {
    float4 v = reinterpret_cast<float4*>(x)[i];
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    reinterpret_cast<float4*>(x)[i] = v;
  }
  if (i < n - 4 * n4) x[4 * n4 + i] *= f;  // tail (n not a multiple of 4)
}

__global__ void table_rnorm_kernel(const float* table, float* out, int64_t rows, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) {
    const float e = table[row * H + c];
    s += e * e;
  }
  s = xf_wave_sum(s);
  if (lane == 0) out[row] = 1.f / fmaxf(sqrtf(s), 1e-8f);
}

int npl_of(int H) { return (H + 63) / 64; }

template <bool GATHER>
int launch_ln_fwd(const LnFwdArgs& a, hipStream_t st) {
  dim3 block(256);
  if (a.H == 64 || a.H == 128 || a.H == 256) {
    const int rpw = 256 / a.H;  // rows per wave
    dim3 gridv((unsigned)((a.rows + 4 * rpw - 1) / (4 * rpw)));
    if (a.H == 64) hipLaunchKernelGGL((ln_fwd_v4_kernel<16, GATHER>), gridv, block, 0, st, a);
    else if (a.H == 128) hipLaunchKernelGGL((ln_fwd_v4_kernel<32, GATHER>), gridv, block, 0, st, a);
    else hipLaunchKernelGGL((ln_fwd_v4_kernel<64, GATHER>), gridv, block, 0, st, a);
    XF_LAUNCH_CHECK();
    return XFMR_OK;
  }
  dim3 grid((unsigned)((a.rows + 3) / 4));
  const int npl = npl_of(a.H);
  if (npl <= 1) hipLaunchKernelGGL((ln_fwd_kernel<1, GATHER>), grid, block, 0, st, a);
  else if (npl <= 2) hipLaunchKernelGGL((ln_fwd_kernel<2, GATHER>), grid, block, 0, st, a);
  else if (npl <= 4) hipLaunchKernelGGL((ln_fwd_kernel<4, GATHER>), grid, block, 0, st, a);
  else if (npl <= 8) hipLaunchKernelGGL((ln_fwd_kernel<8, GATHER>), grid, block, 0, st, a);
  else if (npl <= kMaxPerLane) hipLaunchKernelGGL((ln_fwd_kernel<16, GATHER>), grid, block, 0, st, a);
  else return XFMR_EUNSUPPORTED;
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int ln_bwd_blocks(int64_t rows, int H, int* rows_per_block) {
  // 32 rows per workgroup, <= 1024 workgroups (measured at T = 25 600: 1.859 ms/step; 64 rows / 512: 1.877;
  // 16 rows / 2048: 1.891 -- the partial records the final reduction reads grow with the workgroup count).
  // Small steps (round 4): 32 rows per workgroup are 8 dependent row round trips per wave on rows / 32 CUs -- 1 024 tokens
  // of H = 384 (the reference's default model) took 33 us per LayerNorm backward, a quarter of that step. Below 16 384 rows:
  // 16 per workgroup; below 4 096: one wave-iteration (4 waves x 64 / lanes-per-row rows).
  const int iter_rows = H == 64 ? 16 : H == 128 ? 8 : 4;  // rows of one workgroup iteration (ln_bwd_v4_kernel / ln_bwd_kernel)
  const int64_t per = rows >= 16384 ? 32 : rows >= 4096 ? 16 : iter_rows;
  int64_t blocks = (rows + per - 1) / per;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  int64_t rpb = (rows + blocks - 1) / blocks;
  rpb = ((rpb + iter_rows - 1) / iter_rows) * iter_rows;  // whole workgroup iterations
  *rows_per_block = (int)rpb;
  return (int)((rows + rpb - 1) / rpb);
}
// The most partial records ln_bwd_blocks gives for ANY row count <= rows (the plan is not monotone in the row count; the
// packed layout carves for batch x seq_len rows and launches with the real ones).
int ln_bwd_blocks_bound(int64_t rows) {
  const int64_t b = (rows + 3) / 4;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}


// ---- packed rows (xfmr_encoder_cfg.seq_offsets) ----------------------------------------------------------------------
// one block row per sequence: rows [0, len_b) of the three (B, L) index tensors -> their packed places
// `order` (optional): packed slot b holds batch row order[b] -- the caller sorts the sequences by length, longest first, so
// that the one-workgroup-per-(sequence, head) attention kernels start their longest workgroups first (xfmr_pack_rows_ordered)
__global__ __launch_bounds__(256) void pack_rows_kernel(const int64_t* hist, const int64_t* pos, const int64_t* neg,
                                                        const int64_t* offs64, const int64_t* order, int B, int L,
                                                        int64_t packed_rows, int64_t* hist_p, int64_t* pos_p, int64_t* neg_p,
                                                        int32_t* offs32, int32_t* row_pos) {
  const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
  const int64_t o0 = offs64[b], o1 = offs64[b + 1];
  if (l == 0) {
    offs32[b] = (int32_t)o0;
    if (b == (int)gridDim.y - 1) offs32[b + 1] = (int32_t)o1;
  }
  const int64_t len = o1 - o0;
  if (l >= len || len > L || o0 + l >= packed_rows) return;  // (a malformed offset table writes nothing out of range)
  int64_t brow = b;
  if (order) {
    brow = order[b];
    if (brow < 0 || brow >= B) return;  // (a malformed order reads nothing out of range)
  }
  const int64_t src = brow * L + l, dst = o0 + l;
  hist_p[dst] = hist[src];
  pos_p[dst] = pos[src];
  if (neg) neg_p[dst] = neg[src];
  row_pos[dst] = l;
}
// position-embedding gradient from packed rows: one block of 16 waves per position p; wave w adds the sequences
// b = w, w + 16, ... that are longer than p (loads of eight sequences in flight), the waves are combined through LDS in
// wave order: a fixed order, bit-reproducible
__global__ __launch_bounds__(1024) void pos_grad_packed_kernel(const float* d_pre, float* d_pos, const int32_t* offs, int B,
                                                               int H) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16][H]
  const int p = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float acc[kMaxPerLane];
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) acc[i] = 0.f;
  for (int b0 = w; b0 < B; b0 += 16 * 8) {
    int row[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      row[u] = -1;
      if (b < B) {
        const int o0 = offs[b], len = offs[b + 1] - o0;
        if (p < len) row[u] = o0 + p;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (row[u] < 0) continue;  // (wave-uniform)
#pragma unroll
      for (int i = 0; i < kMaxPerLane; ++i) {
        const int c = lane + 64 * i;
        if (c < H) acc[i] += d_pre[(int64_t)row[u] * H + c];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kMaxPerLane; ++i) {
    const int c = lane + 64 * i;
    if (c < H) red[w * H + c] = acc[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 1024) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q * H + c];
    d_pos[(int64_t)p * H + c] = s;
  }
}
}  // namespace

extern "C" {

int xfmr_embed_ln_fwd(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                      const float* type_emb, const float* gamma, const float* beta, float* out, float* pre,
                      float* mean, float* rstd, uint8_t* key_mask, int32_t B, int32_t L, int32_t H, float eps,
                      float dropout_p, uint64_t seed, uint32_t site, void* stream) {
  return xf_embed_ln_fwd_ex(item_idx, table, n_rows, pos_emb, type_emb, gamma, beta, out, nullptr, pre, mean, rstd,
                            key_mask, B, L, H, eps, dropout_p, seed, site, (hipStream_t)stream);
}

int xf_embed_ln_fwd_ex(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                       const float* type_emb, const float* gamma, const float* beta, float* out, void* out16,
                       float* pre, float* mean, float* rstd, uint8_t* key_mask, int32_t B, int32_t L, int32_t H,
                       float eps, float dropout_p, XfSeed seed, uint32_t site, hipStream_t stream) {
  if (!item_idx || !table || !pos_emb || !type_emb || !gamma || !beta || !out || !pre || !mean || !rstd || !key_mask)
    return XFMR_EINVAL;
  if (B <= 0 || L <= 0 || H <= 0 || n_rows <= 0) return XFMR_EINVAL;
  LnFwdArgs a{};
  a.x = nullptr; a.idx = item_idx; a.table = table; a.n_rows = n_rows; a.pos_emb = pos_emb; a.type_emb = type_emb;
  a.L = L; a.gamma = gamma; a.beta = beta; a.y = out; a.pre = pre; a.mean = mean; a.rstd = rstd;
  a.key_mask = key_mask; a.rows = (int64_t)B * L; a.H = H; a.eps = eps;
  a.drop = xf_make_dropout(dropout_p, seed, site);
  a.y16 = reinterpret_cast<__bf16*>(out16);
  return launch_ln_fwd<true>(a, stream);
}

// packed rows (xfmr_encoder_cfg.seq_offsets): `rows` packed rows, row r adds position row_pos[r]
int xf_embed_ln_fwd_packed_ex(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                              const float* type_emb, const float* gamma, const float* beta, float* out, void* out16,
                              float* pre, float* mean, float* rstd, uint8_t* key_mask, int64_t rows, const int32_t* row_pos,
                              int32_t H, float eps, float dropout_p, XfSeed seed, uint32_t site, hipStream_t stream) {
  if (!item_idx || !table || !pos_emb || !type_emb || !gamma || !beta || !out || !pre || !mean || !rstd || !key_mask || !row_pos)
    return XFMR_EINVAL;
  if (rows <= 0 || H <= 0 || n_rows <= 0) return XFMR_EINVAL;
  LnFwdArgs a{};
  a.x = nullptr; a.idx = item_idx; a.table = table; a.n_rows = n_rows; a.pos_emb = pos_emb; a.type_emb = type_emb;
  a.L = 1; a.row_pos = row_pos; a.gamma = gamma; a.beta = beta; a.y = out; a.pre = pre; a.mean = mean; a.rstd = rstd;
  a.key_mask = key_mask; a.rows = rows; a.H = H; a.eps = eps;
  a.drop = xf_make_dropout(dropout_p, seed, site);
  a.y16 = reinterpret_cast<__bf16*>(out16);
  return launch_ln_fwd<true>(a, stream);
}

int xfmr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t rows, int32_t H, float eps, void* stream) {
  return xf_layernorm_fwd_ex(x, gamma, beta, y, nullptr, mean, rstd, rows, H, eps, (hipStream_t)stream);
}

int xf_layernorm_fwd_ex(const float* x, const float* gamma, const float* beta, float* y, void* y16, float* mean,
                        float* rstd, int64_t rows, int32_t H, float eps, hipStream_t stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || H <= 0) return XFMR_EINVAL;
  LnFwdArgs a{};
  a.x = x; a.gamma = gamma; a.beta = beta; a.y = y; a.mean = mean; a.rstd = rstd; a.rows = rows; a.H = H;
  a.eps = eps; a.L = 1;
  a.drop = xf_make_dropout(0.f, 0, 0);
  a.y16 = reinterpret_cast<__bf16*>(y16);
  return launch_ln_fwd<false>(a, stream);
}

size_t xfmr_layernorm_bwd_workspace(int64_t rows, int32_t H) {
  return (size_t)ln_bwd_blocks_bound(rows) * 3 * (size_t)H * sizeof(float);  // enough for every row count <= rows
}

// Internal variant that also takes the dropout applied to the LayerNorm OUTPUT (embedding site).
int xf_layernorm_bwd_impl(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          float* dx, void* d_lin, bool lin16, float* d_gamma, float* d_beta, float* d_bias,
                          int64_t rows, int32_t H, XfDropout drop_out, XfDropout drop_lin, void* partials,
                          hipStream_t st, int* blocks_out) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !partials || rows <= 0 || H <= 0) return XFMR_EINVAL;
  if (drop_lin.on && !d_lin) return XFMR_EINVAL;
  LnBwdArgs a{};
  a.dy = dy; a.x = x; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.dx = dx; a.d_lin = d_lin;
  a.partials = (float*)partials; a.rows = rows; a.H = H; a.lin16 = lin16 ? 1 : 0;
  const int blocks = ln_bwd_blocks(rows, H, &a.rows_per_block);
  a.drop_out = drop_out; a.drop_lin = drop_lin;
  const size_t shmem = (size_t)12 * H * sizeof(float);
  const int npl = npl_of(H);
  dim3 grid(blocks), block(256);
  if (H == 64) hipLaunchKernelGGL((ln_bwd_v4_kernel<16>), grid, block, 0, st, a);
  else if (H == 128) hipLaunchKernelGGL((ln_bwd_v4_kernel<32>), grid, block, 0, st, a);
  else if (H == 256) hipLaunchKernelGGL((ln_bwd_v4_kernel<64>), grid, block, 0, st, a);
  else if (npl <= 1) hipLaunchKernelGGL((ln_bwd_kernel<1>), grid, block, shmem, st, a);
  else if (npl <= 2) hipLaunchKernelGGL((ln_bwd_kernel<2>), grid, block, shmem, st, a);
  else if (npl <= 4) hipLaunchKernelGGL((ln_bwd_kernel<4>), grid, block, shmem, st, a);
  else if (npl <= 8) hipLaunchKernelGGL((ln_bwd_kernel<8>), grid, block, shmem, st, a);
  else if (npl <= kMaxPerLane) hipLaunchKernelGGL((ln_bwd_kernel<16>), grid, block, shmem, st, a);
  else return XFMR_EUNSUPPORTED;
  XF_LAUNCH_CHECK();
  if (blocks_out) *blocks_out = blocks;
  if (!d_gamma && !d_beta && !d_bias) return XFMR_OK;  // reduction deferred to the caller
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((3 * H + 63) / 64), dim3(1024), 0, st, (const float*)partials,
                     blocks, H, d_gamma, d_beta, d_bias);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* d_lin, float* d_gamma, float* d_beta, float* d_bias, int64_t rows,
                       int32_t H, float dropout_p, uint64_t seed, uint32_t site, void* partials, void* stream) {
  // public contract: d_lin is written only when dropout is on (otherwise it equals dx)
  return xf_layernorm_bwd_impl(dy, x, mean, rstd, gamma, dx, dropout_p > 0.f ? d_lin : nullptr, false, d_gamma, d_beta,
                               d_bias, rows, H, xf_make_dropout(0.f, 0, 0), xf_make_dropout(dropout_p, seed, site), partials,
                               (hipStream_t)stream);
}

int xfmr_embed_param_grads(const float* d_pre, float* d_pos, float* d_type, int32_t B, int32_t L, int32_t H,
                           int32_t max_pos, void* stream) {
  if (!d_pre || !d_pos || !d_type || B <= 0 || L <= 0 || H <= 0 || max_pos < L) return XFMR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // d_pos (L*H) = column sums of d_pre viewed as [B][L*H]; rows L..max_pos-1 get no gradient
  if (int rc = xf_rowsum(d_pos, d_pre, B, (int64_t)L * H, st)) return rc;
  if (max_pos > L &&
      xf_zero_async(d_pos + (int64_t)L * H, (size_t)(max_pos - L) * H * sizeof(float), st) != hipSuccess)
    return XFMR_EHIP;
  // d_type[0] = column sums of d_pos viewed as [L][H]; d_type[1] = 0 (token type 1 is never used)
  if (int rc = xf_rowsum(d_type, d_pos, L, H, st)) return rc;
  if (xf_zero_async(d_type + H, (size_t)H * sizeof(float), st) != hipSuccess) return XFMR_EHIP;
  return XFMR_OK;
}

// packed rows: d_pos[p] = sum over the sequences longer than p of d_pre[offs[b] + p] (ascending b: a fixed order)
int xf_embed_param_grads_packed(const float* d_pre, float* d_pos, float* d_type, const int32_t* offs, int32_t B, int32_t L,
                                int32_t H, int32_t max_pos, hipStream_t st) {
  if (!d_pre || !d_pos || !d_type || !offs || B <= 0 || L <= 0 || H <= 0 || max_pos < L) return XFMR_EINVAL;
  if (H > 64 * kMaxPerLane) return XFMR_EUNSUPPORTED;
  hipLaunchKernelGGL(pos_grad_packed_kernel, dim3((unsigned)max_pos), dim3(1024), (size_t)16 * H * sizeof(float), st, d_pre,
                     d_pos, offs, B, H);
  XF_LAUNCH_CHECK();
  if (int rc = xf_rowsum(d_type, d_pos, L, H, st)) return rc;
  if (xf_zero_async(d_type + H, (size_t)H * sizeof(float), st) != hipSuccess) return XFMR_EHIP;
  return XFMR_OK;
}

int xfmr_pack_rows_ordered(const int64_t* hist, const int64_t* pos, const int64_t* neg, const int64_t* offsets64,
                           const int64_t* order, int32_t batch, int32_t seq_len, int64_t packed_rows, int64_t* hist_p,
                           int64_t* pos_p, int64_t* neg_p, int32_t* offsets32, int32_t* row_pos, void* stream) {
  if (!hist || !pos || !offsets64 || !hist_p || !pos_p || !offsets32 || !row_pos) return XFMR_EINVAL;
  if ((neg == nullptr) != (neg_p == nullptr)) return XFMR_EINVAL;
  if (batch <= 0 || seq_len <= 0 || packed_rows < 0 || packed_rows > (int64_t)batch * seq_len) return XFMR_EINVAL;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)((seq_len + 255) / 256), (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, hist, pos, neg, offsets64, order, batch, seq_len, packed_rows, hist_p, pos_p, neg_p,
                     offsets32, row_pos);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
int xfmr_pack_rows(const int64_t* hist, const int64_t* pos, const int64_t* neg, const int64_t* offsets64, int32_t batch,
                   int32_t seq_len, int64_t packed_rows, int64_t* hist_p, int64_t* pos_p, int64_t* neg_p,
                   int32_t* offsets32, int32_t* row_pos, void* stream) {
  return xfmr_pack_rows_ordered(hist, pos, neg, offsets64, nullptr, batch, seq_len, packed_rows, hist_p, pos_p, neg_p,
                                offsets32, row_pos, stream);
}

int xfmr_pool(const float* tok, const uint8_t* key_mask, float* out, int32_t B, int32_t L, int32_t H, int32_t mode,
              void* stream) {
  if (!tok || !key_mask || !out || B <= 0 || L <= 0 || H <= 0) return XFMR_EINVAL;
  if (mode < XFMR_POOL_MEAN || mode > XFMR_POOL_LASTTOKEN) return XFMR_EINVAL;
  hipLaunchKernelGGL(pool_kernel, dim3((H + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, tok, key_mask, out, L, H,
                     mode);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
int xfmr_mean_pool(const float* tok, const uint8_t* key_mask, float* out, int32_t B, int32_t L, int32_t H,
                   void* stream) {
  return xfmr_pool(tok, key_mask, out, B, L, H, XFMR_POOL_MEAN, stream);
}

int xfmr_l2_normalize_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int32_t H, float eps, void* stream) {
  if (!x || !y || rows <= 0 || H <= 0 || !(eps > 0.f)) return XFMR_EINVAL;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                     inv_norm, rows, H, eps);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
int xfmr_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int32_t H,
                          float eps, void* stream) {
  if (!dy || !y || !inv_norm || !dx || rows <= 0 || H <= 0 || !(eps > 0.f)) return XFMR_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, y,
                     inv_norm, dx, rows, H, eps);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
               float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
               void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return XFMR_EINVAL;
  if (!xf_aligned16(params) || !xf_aligned16(grads) || !xf_aligned16(exp_avg) || !xf_aligned16(exp_avg_sq))
    return XFMR_EALIGN;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, (float)bc1,
                     (float)sqrt(bc2), grad_scale, (const uint32_t*)nullptr, 0);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_adamw_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, const uint32_t* step_device,
                   int32_t step_offset, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || !step_device) return XFMR_EINVAL;
  if (!xf_aligned16(params) || !xf_aligned16(grads) || !xf_aligned16(exp_avg) || !xf_aligned16(exp_avg_sq))
    return XFMR_EALIGN;
  const int64_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale,
                     step_device, (int)step_offset);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

__global__ void step_advance_kernel(uint32_t* c) { *c += 1; }
int xfmr_step_advance(uint32_t* step_device, void* stream) {
  if (!step_device) return XFMR_EINVAL;
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_device);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_scale_by_device_scalar(float* x, int64_t n, const float* scalar, void* stream) {
  if (!x || !scalar || n <= 0) return XFMR_EINVAL;
  const int64_t n4 = xf_aligned16(x) ? n / 4 : 0;
  const int64_t work = n4 > 0 ? (n4 > n - 4 * n4 ? n4 : n - 4 * n4) : n;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n4, n,
                     scalar);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

int xfmr_table_rnorm(const float* table, float* table_rnorm, int64_t n_rows, int32_t H, void* stream) {
  if (!table || !table_rnorm || n_rows <= 0 || H <= 0) return XFMR_EINVAL;
  hipLaunchKernelGGL(table_rnorm_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     table, table_rnorm, n_rows, H);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

}  // extern "C"
