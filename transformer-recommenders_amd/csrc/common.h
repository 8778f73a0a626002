// Shared device helpers for libxfmr_hip (gfx950 / CDNA4 only: wave64, MFMA 32x32, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xfmr_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define XF_WAVE 64

#define XF_LAUNCH_CHECK()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return XFMR_EHIP;  \
  } while (0)

static inline bool xf_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ int xf_lane() { return threadIdx.x & 63; }

// 32x32 MFMA accumulator map (dtype independent on gfx950): element r of lane l is
// C[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ int xf_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------------
// Precision policies. A policy names the LDS element type and implements, for one wave,
//   tile_nt : acc(32x32) += A[arow0..+32][0..K) * B[brow0..+32][0..K)^T   (both images K-contiguous)
//   tile_xb : acc(32x32) += A[arow0..+32][kbase + k] * X[k][col]  for k = 0..31, where X is a 32x32
//             accumulator tile still in registers (its ROW index is the contraction index: no lane
//             movement, cdna_hip_programming.md section 3 "An accumulator tile as the next MFMA's
//             operand"); A is an image whose K runs over X's rows.
// Images are row-major [row][ld]; ld is in elements and keeps 16-byte row alignment.
// ---------------------------------------------------------------------------------------------------
struct PrecBF16 {
  using elem = __bf16;
  static constexpr int kPad = 8;  // elements of row padding (16 B = one ds_read_b128)
  static constexpr int kId = XFMR_PREC_BF16;
  __device__ static __forceinline__ elem cvt(float x) { return (__bf16)x; }
  __device__ static __forceinline__ float round(float x) { return (float)((__bf16)x); }

  __device__ static __forceinline__ void tile_nt(f32x16& acc, const elem* A, int lda, int arow0, const elem* B,
                                                 int ldb, int brow0, int K) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + 8 * (l >> 5);
    const elem* pb = B + (brow0 + (l & 31)) * ldb + 8 * (l >> 5);
#pragma unroll 4
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(pa + k0);
      bf16x8 b = *reinterpret_cast<const bf16x8*>(pb + k0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  // B operand held in registers (8 bf16 per 16-deep k-step), e.g. a wave's query rows.
  __device__ static __forceinline__ void tile_nreg(f32x16& acc, const elem* A, int lda, int arow0,
                                                   const bf16x8* breg, int K) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + 8 * (l >> 5);
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      bf16x8 a = *reinterpret_cast<const bf16x8*>(pa + 16 * s);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, breg[s], acc, 0, 0, 0);
    }
  }
  __device__ static __forceinline__ void tile_xb(f32x16& acc, const elem* A, int lda, int arow0, int kbase,
                                                 const f32x16& x) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + kbase + 4 * (l >> 5);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[8 * s + j];
      // element j of lane half h pairs with X row 16s + 8(j>>2) + 4h + (j&3)
      bf16x4 a0 = *reinterpret_cast<const bf16x4*>(pa + 16 * s);
      bf16x4 a1 = *reinterpret_cast<const bf16x4*>(pa + 16 * s + 8);
      bf16x8 a;
      a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3];
      a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
};

struct PrecF32 {
  using elem = float;
  static constexpr int kPad = 4;
  static constexpr int kId = XFMR_PREC_F32;
  __device__ static __forceinline__ elem cvt(float x) { return x; }
  __device__ static __forceinline__ float round(float x) { return x; }

  __device__ static __forceinline__ void tile_nt(f32x16& acc, const elem* A, int lda, int arow0, const elem* B,
                                                 int ldb, int brow0, int K) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + (l >> 5);
    const elem* pb = B + (brow0 + (l & 31)) * ldb + (l >> 5);
#pragma unroll 8
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[k0], pb[k0], acc, 0, 0, 0);
  }
  // breg[s] holds B[k = 2s + (lane>>5)][col = lane&31]
  __device__ static __forceinline__ void tile_nreg(f32x16& acc, const elem* A, int lda, int arow0, const float* breg,
                                                   int K) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + (l >> 5);
#pragma unroll
    for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * s], breg[s], acc, 0, 0, 0);
  }
  __device__ static __forceinline__ void tile_xb(f32x16& acc, const elem* A, int lda, int arow0, int kbase,
                                                 const f32x16& x) {
    const int l = xf_lane();
    const elem* pa = A + (arow0 + (l & 31)) * lda + kbase + 4 * (l >> 5);
    // k-step r contracts X rows (r&3)+8(r>>2) [lane half 0] and +4 [lane half 1]: exactly register r.
#pragma unroll
    for (int r = 0; r < 16; ++r)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[(r & 3) + 8 * (r >> 2)], x[r], acc, 0, 0, 0);
  }
};

template <class P>
__host__ __device__ constexpr int xf_ld(int k) {  // padded leading dimension of a [rows][k] image
  return k + P::kPad;
}

// Register-resident "N" operand of one wave (its 32 rows, K deep), loaded straight from global fp32.
template <class P, int K>
struct RegRows;
template <int K>
struct RegRows<PrecBF16, K> {
  bf16x8 v[K / 16];
  // row pointer of THIS lane's row (lane&31); valid==false loads zeros
  __device__ __forceinline__ void load(const float* row, bool valid) {
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      float4 a = make_float4(0, 0, 0, 0), b = a;
      if (valid) {
        a = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h);
        b = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h + 4);
      }
      v[s][0] = (__bf16)a.x; v[s][1] = (__bf16)a.y; v[s][2] = (__bf16)a.z; v[s][3] = (__bf16)a.w;
      v[s][4] = (__bf16)b.x; v[s][5] = (__bf16)b.y; v[s][6] = (__bf16)b.z; v[s][7] = (__bf16)b.w;
    }
  }
  // only the first n columns of the row exist (n a multiple of 16; the rest read as zero): a row narrower than K
  __device__ __forceinline__ void load_n(const float* row, bool valid, int n) {
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      float4 a = make_float4(0, 0, 0, 0), b = a;
      if (valid && 16 * s < n) {
        a = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h);
        b = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h + 4);
      }
      v[s][0] = (__bf16)a.x; v[s][1] = (__bf16)a.y; v[s][2] = (__bf16)a.z; v[s][3] = (__bf16)a.w;
      v[s][4] = (__bf16)b.x; v[s][5] = (__bf16)b.y; v[s][6] = (__bf16)b.z; v[s][7] = (__bf16)b.w;
    }
  }
  // `row` must be dereferenceable even when !valid (the result is then zero): every load is issued up front,
  // without the per-piece branches (and the chain of exposed load latencies) of load()
  __device__ __forceinline__ void load_safe(const float* row, bool valid) {
    const int h = xf_lane() >> 5;
    float4 a[K / 16], b[K / 16];
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      a[s] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h);
      b[s] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h + 4);
    }
    const float z = valid ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      v[s][0] = (__bf16)(a[s].x * z); v[s][1] = (__bf16)(a[s].y * z); v[s][2] = (__bf16)(a[s].z * z);
      v[s][3] = (__bf16)(a[s].w * z); v[s][4] = (__bf16)(b[s].x * z); v[s][5] = (__bf16)(b[s].y * z);
      v[s][6] = (__bf16)(b[s].z * z); v[s][7] = (__bf16)(b[s].w * z);
    }
  }
  // load_safe + the sum of squares of THIS lane's fp32 pieces (before the bf16 rounding): the row is read once for the
  // operand and for its norm (the lane pair l, l ^ 32 covers the whole row between them)
  __device__ __forceinline__ float load_safe_sq(const float* row, bool valid) {
    const int h = xf_lane() >> 5;
    float4 a[K / 16], b[K / 16];
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      a[s] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h);
      b[s] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * h + 4);
    }
    const float z = valid ? 1.f : 0.f;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      sq += a[s].x * a[s].x + a[s].y * a[s].y + a[s].z * a[s].z + a[s].w * a[s].w;
      sq += b[s].x * b[s].x + b[s].y * b[s].y + b[s].z * b[s].z + b[s].w * b[s].w;
      v[s][0] = (__bf16)(a[s].x * z); v[s][1] = (__bf16)(a[s].y * z); v[s][2] = (__bf16)(a[s].z * z);
      v[s][3] = (__bf16)(a[s].w * z); v[s][4] = (__bf16)(b[s].x * z); v[s][5] = (__bf16)(b[s].y * z);
      v[s][6] = (__bf16)(b[s].z * z); v[s][7] = (__bf16)(b[s].w * z);
    }
    return sq * z;
  }
  __device__ __forceinline__ void load(const __bf16* row, bool valid) {  // bf16 storage: no conversion
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) {
      if (valid) v[s] = *reinterpret_cast<const bf16x8*>(row + 16 * s + 8 * h);
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] = (__bf16)0.f;
    }
  }
  template <bool S16>
  __device__ __forceinline__ void load_at(const void* base, int64_t idx, bool valid) {
    if (S16) load(reinterpret_cast<const __bf16*>(base) + idx, valid);
    else load(reinterpret_cast<const float*>(base) + idx, valid);
  }
  __device__ __forceinline__ float dot_partial(const RegRows& o) const {  // over this lane's half of K
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < K / 16; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf((float)v[s][j], (float)o.v[s][j], acc);
    return acc;
  }
  __device__ __forceinline__ const bf16x8* regs() const { return v; }
  // write this lane's part of its row into a K-contiguous image row (for tile_nt)
  __device__ __forceinline__ void store_image(__bf16* img_row) const {
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 16; ++s) *reinterpret_cast<bf16x8*>(img_row + 16 * s + 8 * h) = v[s];
  }
};
template <int K>
struct RegRows<PrecF32, K> {
  float v[K / 2];
  __device__ __forceinline__ void load(const float* row, bool valid) {
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 2; ++s) v[s] = valid ? row[2 * s + h] : 0.f;
  }
  __device__ __forceinline__ void load_n(const float* row, bool valid, int n) {  // columns >= n read as zero
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 2; ++s) v[s] = (valid && 2 * s < n) ? row[2 * s + h] : 0.f;
  }
  __device__ __forceinline__ float dot_partial(const RegRows& o) const {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < K / 2; ++s) acc = fmaf(v[s], o.v[s], acc);
    return acc;
  }
  __device__ __forceinline__ const float* regs() const { return v; }
  __device__ __forceinline__ void store_image(float* img_row) const {
    const int h = xf_lane() >> 5;
#pragma unroll
    for (int s = 0; s < K / 2; ++s) img_row[2 * s + h] = v[s];
  }
};

// ---------------------------------------------------------------------------------------------------
// Activation storage. Tensors that are only ever consumed as MFMA operands of the bf16 policy (qkv, ctx,
// gelu output, the gradients flowing into dX / dW GEMMs) may live in HBM as bf16 instead of fp32: the
// consumer rounds to bf16 on the way into LDS anyway, so the products are bit-identical and the bytes halve.
// S16 = true selects that storage; pointers are type-erased at the launch boundary.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 xf_bf16x4_to_f32(uint2 raw) {
  float4 v;
  v.x = __uint_as_float(raw.x << 16); v.y = __uint_as_float(raw.x & 0xffff0000u);
  v.z = __uint_as_float(raw.y << 16); v.w = __uint_as_float(raw.y & 0xffff0000u);
  return v;
}
__device__ __forceinline__ uint2 xf_f32x4_to_bf16(float4 v) {
  union { bf16x4 b; uint2 u; } o;
  o.b[0] = (__bf16)v.x; o.b[1] = (__bf16)v.y; o.b[2] = (__bf16)v.z; o.b[3] = (__bf16)v.w;
  return o.u;
}
// 4 consecutive elements starting at element index idx (idx % 4 == 0)
template <bool S16>
__device__ __forceinline__ float4 xf_ld4(const void* base, int64_t idx) {
  if (S16) return xf_bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(base) + idx));
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
template <bool S16>
__device__ __forceinline__ void xf_st4(void* base, int64_t idx, float4 v) {
  if (S16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(base) + idx) = xf_f32x4_to_bf16(v);
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}
template <bool S16>
__device__ __forceinline__ const void* xf_at(const void* base, int64_t idx) {
  return S16 ? (const void*)(reinterpret_cast<const __bf16*>(base) + idx)
             : (const void*)(reinterpret_cast<const float*>(base) + idx);
}
template <bool S16>
__device__ __forceinline__ void* xf_at(void* base, int64_t idx) {
  return S16 ? (void*)(reinterpret_cast<__bf16*>(base) + idx) : (void*)(reinterpret_cast<float*>(base) + idx);
}

// ---------------------------------------------------------------------------------------------------
// LDS staging from fp32 global memory, converting to the policy's element type.
// ---------------------------------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ void xf_store4(typename P::elem* dst, float4 v);
template <>
__device__ __forceinline__ void xf_store4<PrecBF16>(__bf16* dst, float4 v) {
  bf16x4 o;
  o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4*>(dst) = o;
}
template <>
__device__ __forceinline__ void xf_store4<PrecF32>(float* dst, float4 v) {
  *reinterpret_cast<float4*>(dst) = v;
}
template <class P>
__device__ __forceinline__ void xf_store2(typename P::elem* dst, float a, float b);
template <>
__device__ __forceinline__ void xf_store2<PrecBF16>(__bf16* dst, float a, float b) {
  bf16x2 o;
  o[0] = (__bf16)a; o[1] = (__bf16)b;
  *reinterpret_cast<bf16x2*>(dst) = o;
}
template <>
__device__ __forceinline__ void xf_store2<PrecF32>(float* dst, float a, float b) {
  *reinterpret_cast<float2*>(dst) = make_float2(a, b);
}

__device__ __forceinline__ float xf_get(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }

// Wave priority of the training chain's kernels (experiment switch -DXF_CHAIN_SETPRIO=n, n = 1..3; default off): the
// low-priority side streams only decide which WORKGROUP is dispatched first -- once resident, a logging-pass or dW wave
// arbitrates for issue slots like any other. s_setprio raises the chain's waves above them on a shared SIMD.
#ifdef XF_CHAIN_SETPRIO
#define XF_CHAIN_PRIO() __builtin_amdgcn_s_setprio(XF_CHAIN_SETPRIO)
#else
#define XF_CHAIN_PRIO() ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------
// Dropout: stateless per-element hash so that forward and backward kernels with different thread
// mappings regenerate the same mask. keep  <=>  hash32(element ^ key) >= threshold(p).
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t xf_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// The dropout stream of a step: the host-side seed, optionally mixed ON THE DEVICE with a step counter read from HBM
// (xfmr_encoder_cfg.step_device) -- a captured hipGraph replays the same kernel arguments every step, so whatever
// changes from step to step has to come from device memory. dyn == nullptr: the seed alone (eager launches).
struct XfSeed {
  uint64_t seed;
  const uint32_t* dyn;
  XfSeed(uint64_t s = 0, const uint32_t* d = nullptr) : seed(s), dyn(d) {}
};
struct XfDropout {
  uint32_t key;
  uint32_t thresh;  // drop when hash < thresh
  float scale;      // 1/(1-p); 1 when disabled
  bool on;
  const uint32_t* dyn;  // device step counter still to be mixed into `key` (xf_drop_resolve at kernel entry); or null
};
static inline XfDropout xf_make_dropout(float p, XfSeed sd, uint32_t site) {
  XfDropout d;
  const uint64_t seed = sd.seed;
  d.on = p > 0.f;
  d.key = xf_hash32((uint32_t)seed ^ xf_hash32((uint32_t)(seed >> 32) + 0x9E3779B9U * (site + 1)));
  double t = (double)p * 4294967296.0;
  d.thresh = p >= 1.f ? 0xFFFFFFFFu : (uint32_t)t;
  d.scale = d.on ? 1.f / (1.f - p) : 1.f;
  d.dyn = d.on ? sd.dyn : nullptr;
  return d;
}
// Kernel entry: fold the device-side step counter into the key (one scalar load + a scalar hash per wave; forward and
// backward kernels of a step read the same counter value -- it advances after the optimizer, xfmr_step_advance).
__host__ __device__ __forceinline__ XfDropout xf_drop_resolve(XfDropout d) {
  if (d.dyn) {
    d.key = xf_hash32(d.key ^ (*d.dyn * 0x9E3779B9U + 0x7F4A7C15U));
    d.dyn = nullptr;
  }
  return d;
}
// Two-index form (attention probabilities: query row, key column; hidden states: token row, feature column): the full hash is
// spent once per ROW (xf_drop_rowkey) and an element costs one xor + one multiply + the threshold test on
// (rowkey ^ col * kDropColMul) * C -- the high bits of that product, which decide the comparison, depend on every
// bit of both indices. Keep rate, row / column rates, neighbour correlations and 4-pattern chi-squares measured the
// same as for the per-element hash; the per-element hash was ~1/5 of the attention kernels' time.
constexpr uint32_t kDropColMul = 0x9E3779B1U;
__host__ __device__ __forceinline__ uint32_t xf_drop_rowkey(const XfDropout& d, uint32_t row) {
  return xf_hash32(row ^ d.key);
}
__host__ __device__ __forceinline__ float xf_keep_scale_rc(const XfDropout& d, uint32_t rowkey, uint32_t colmix) {
  return ((rowkey ^ colmix) * 0x7feb352dU >= d.thresh) ? d.scale : 0.f;
}
// Hidden-state dropout (embedding output, attention-output / FFN-output Linears): element (row, col) of a [rows][N]
// tensor. Every kernel that applies or differentiates one dropout site goes through these two, so the masks agree.
// (Until round 2 these sites spent the full 32-bit hash per ELEMENT: 11 vector instructions, two of them quarter-rate
// multiplies, on 234 M elements per step in epilogues that are sensitive to their VALU work.)
__device__ __forceinline__ float xf_keep_scale_2d(const XfDropout& d, uint32_t row, uint32_t col) {
  return xf_keep_scale_rc(d, xf_drop_rowkey(d, row), col * kDropColMul);
}
__device__ __forceinline__ void xf_drop4(const XfDropout& d, uint32_t row, uint32_t col, float4& v) {  // cols col .. col + 3
  const uint32_t rk = xf_drop_rowkey(d, row), cm = col * kDropColMul;
  v.x *= xf_keep_scale_rc(d, rk, cm); v.y *= xf_keep_scale_rc(d, rk, cm + kDropColMul);
  v.z *= xf_keep_scale_rc(d, rk, cm + 2 * kDropColMul); v.w *= xf_keep_scale_rc(d, rk, cm + 3 * kDropColMul);
}

// Zero-fill as a KERNEL launch (bytes % 4 == 0, p 4-byte aligned): hipMemsetAsync nodes captured into a hipGraph were
// not re-executed by later replays on this runtime (ROCm 7.x with torch 2.10: the multiplicity histogram of the loss
// doubled from the second replay on -- scripts/probe/graph_debug.py), and the training step must replay as a graph.
static __global__ __launch_bounds__(256) void xf_zero_kernel(uint32_t* p, int64_t n16, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) reinterpret_cast<uint4*>(p)[i] = make_uint4(0, 0, 0, 0);
  else if (i - n16 < n4) p[4 * n16 + (i - n16)] = 0;
}
static inline hipError_t xf_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return hipSuccess;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3)) return hipMemsetAsync(p, 0, bytes, st);
  const bool a16 = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const int64_t n16 = a16 ? (int64_t)(bytes / 16) : 0, n4 = (int64_t)(bytes / 4) - 4 * n16, work = n16 + n4;
  hipLaunchKernelGGL(xf_zero_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n16, n4);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// wave reductions (wave64)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float xf_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float xf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// Sum over the 16 lanes of a DPP row (lanes 16 i .. 16 i + 15), result in every lane, through data-parallel-primitive
// operands of the adds themselves: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_ror:4, row_ror:8. (__shfl_xor compiles to
// ds_bpermute_b32 -- an LDS-crossbar instruction plus a separate add per step; the LayerNorm epilogues do 16-22 such
// reductions per tile and are bound by instruction issue.)
__device__ __forceinline__ float xf_row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));
  return x;
}
__device__ __forceinline__ float xf_half_swap(float v) { return __shfl_xor(v, 32, 64); }  // lane <-> lane^32

// raw transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32: 1 ulp, no denormal fix-up code around them)
__device__ __forceinline__ float xf_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float xf_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float xf_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// Exact-erf GELU (hidden_act = "gelu", TF:activations.py) and its derivative. erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. fp32 rounding level; measured against fp64: gelu 4.6e-7, gelu' 3.0e-7 absolute -- torch's
// own fp32 gelu is 1.2e-6 off fp64): one v_rcp, one v_exp and eight fma, branch-free, instead of libm's erff. The
// GELU epilogues of the two FFN GEMMs evaluate this on T x I elements per layer and were VALU-bound on erff.
// h(x) = 0.5 - 0.5 erfc(|x| / sqrt 2) = Phi(|x|) - 0.5, from the same A-S 7.1.26 terms with the 1 / sqrt 2 folded into the
// rational's constant and the 0.5 into the polynomial: 0.5 erfc(z) = (0.5 a1 t + ...) t exp(-z^2), t = 1 / (1 + p z),
// z = |x| / sqrt 2. e_out = exp(-x^2 / 2). Then gelu(x) = x Phi(x) = 0.5 x + |x| h(x) and Phi(x) = 0.5 + copysign(h, x):
// two instructions fewer per element than erf -> 1 + erf -> 0.5 x (...), no maximum (its operand would be canonicalised
// first) -- round 3: the fused FFN kernels are bound by their vector instructions.
__device__ __forceinline__ float xf_phi_minus_half(float ax, float x, float& e_out) {  // ax = |x|
  const float t = xf_rcp(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
  float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  p *= t;
  const float e = xf_exp2((x * x) * (-0.5f * 1.4426950408889634f));
  e_out = e;
  return fmaf(-p, e, 0.5f);
}
__device__ __forceinline__ float xf_gelu(float x) {
  float e;
  const float ax = fabsf(x);
  return fmaf(ax, xf_phi_minus_half(ax, x, e), 0.5f * x);
}
// gelu(x) and gelu'(x) = Phi(x) + x phi(x) from ONE evaluation
__device__ __forceinline__ float xf_gelu_both(float x, float& grad) {
  float e;
  const float ax = fabsf(x);
  const float h = xf_phi_minus_half(ax, x, e);
  grad = fmaf(x * 0.3989422804014327f, e, 0.5f + copysignf(h, x));
  return fmaf(ax, h, 0.5f * x);
}
__device__ __forceinline__ float xf_gelu_grad(float x) {
  float e;
  const float h = xf_phi_minus_half(fabsf(x), x, e);
  return fmaf(x * 0.3989422804014327f, e, 0.5f + copysignf(h, x));
}
__device__ __forceinline__ float xf_softplus(float x) {
  return fmaxf(x, 0.f) + 0.6931471805599453f * xf_log2(1.f + xf_exp2(-fabsf(x) * 1.4426950408889634f));
}
__device__ __forceinline__ float xf_sigmoid(float x) {
  const float t = xf_exp2(-fabsf(x) * 1.4426950408889634f);
  return xf_rcp(1.f + t) * (x >= 0.f ? 1.f : t);
}

// Write a wave's 32(d) x 32(row) accumulator tile, transposed, as 32 rows of 32 contiguous floats.
// scratch: per-wave [32][33] floats. dst row pointer for lane-row j: base + (row0 + j) * stride.
__device__ __forceinline__ void xf_store_tile_T(float* scratch, const f32x16& acc, float mul, float* base, int64_t stride,
                                             int row0, int row_end) {
  const int lane = xf_lane();
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[(lane & 31) * 33 + xf_acc_row(r, lane)] = acc[r] * mul;
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave re-reads its own writes
  __builtin_amdgcn_wave_barrier();
  const int j = lane >> 1, half = lane & 1;
  if (row0 + j < row_end) {
    float* dst = base + (int64_t)(row0 + j) * stride + 16 * half;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float4 v;
      v.x = scratch[j * 33 + 16 * half + 4 * u + 0];
      v.y = scratch[j * 33 + 16 * half + 4 * u + 1];
      v.z = scratch[j * 33 + 16 * half + 4 * u + 2];
      v.w = scratch[j * 33 + 16 * half + 4 * u + 3];
      *reinterpret_cast<float4*>(dst + 4 * u) = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// same, to bf16 storage (two 16-byte stores per lane)
__device__ __forceinline__ void xf_store_tile_T(float* scratch, const f32x16& acc, float mul, __bf16* base,
                                                int64_t stride, int row0, int row_end) {
  const int lane = xf_lane();
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[(lane & 31) * 33 + xf_acc_row(r, lane)] = acc[r] * mul;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const int j = lane >> 1, half = lane & 1;
  if (row0 + j < row_end) {
    __bf16* dst = base + (int64_t)(row0 + j) * stride + 16 * half;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)scratch[j * 33 + 16 * half + 8 * u + e];
      *reinterpret_cast<bf16x8*>(dst + 8 * u) = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
}
template <bool S16>
__device__ __forceinline__ void xf_store_tile_T_at(float* scratch, const f32x16& acc, float mul, void* base,
                                                   int64_t off, int64_t stride, int row0, int row_end) {
  if (S16) xf_store_tile_T(scratch, acc, mul, reinterpret_cast<__bf16*>(base) + off, stride, row0, row_end);
  else xf_store_tile_T(scratch, acc, mul, reinterpret_cast<float*>(base) + off, stride, row0, row_end);
}


// ---------------------------------------------------------------------------------------------------
// Swizzled bf16 row images: what an LDS-DMA gather (global_load_lds_dwordx4, one 1-KiB wave instruction =
// 64 lanes x 16 B, destination = uniform base + lane*16) can produce. Rows are UNPADDED (H*2 bytes), so the
// bank-conflict fix is an XOR swizzle applied on the SOURCE address at gather time and on every read:
// logical 16-byte chunk c of row r lives at chunk position c ^ swz(r).
//   * row-operand fragments (ds_read_b128, 16 lanes = 16 rows): positions c ^ r are 16 distinct slots
//   * transposed fragments for "contract over the image's ROW index" products come from
//     ds_read_b64_tr_b16 on the SAME image (no second, transposed copy of the tile).
// ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short xf_s16x4;

template <int H>
struct SwzImg {
  static constexpr int CPR = H / 8;                         // 16-byte chunks per row
  // Swizzle term of row r: a bijection of the row's low bits that puts bits 0-1 of the row into bits 2-3 of
  // the chunk position. Row-operand reads (16 lanes = 16 consecutive rows, same logical chunk) then hit 16
  // distinct 16-byte slots, and transposed reads (4 consecutive rows x 4 consecutive chunks per half-wave)
  // hit four different chunk groups instead of the same one (measured: 53 % of LDS cycles were conflicts with
  // the plain c ^ (r & 15) swizzle).
  __device__ static __forceinline__ int swz(int r) {
    if (CPR == 4) return (r >> 2) & 3;  // 64-byte rows (attention heads): four rows per 256-B bank row
    if (CPR >= 16) return ((r & 3) << 2) | ((r >> 2) & 3);
    return (((r >> 1) & 1) << 2) | ((r & 1) << 1) | ((r >> 2) & 1);  // CPR == 8: two rows per 256-B bank row
  }
  __device__ static __forceinline__ int off(int r, int c) { return r * H + 8 * (c ^ swz(r)); }

  // acc += A[arow0..+32][0..H) * B^T, B rows in registers (8 bf16 per 16-deep k-step)
  __device__ static __forceinline__ void tile_nreg(f32x16& acc, const __bf16* A, int arow0, const bf16x8* breg) {
    const int l = xf_lane(), r = arow0 + (l & 31), hh = l >> 5;
#pragma unroll
    for (int s = 0; s < H / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + off(r, 2 * s + hh));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, breg[s], acc, 0, 0, 0);
    }
  }
  // acc += A[arow0..+32] * B[brow0..+32]^T, both swizzled images
  __device__ static __forceinline__ void tile_nt(f32x16& acc, const __bf16* A, int arow0, const __bf16* B, int brow0) {
    const int l = xf_lane(), ra = arow0 + (l & 31), rb = brow0 + (l & 31), hh = l >> 5;
#pragma unroll
    for (int s = 0; s < H / 16; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + off(ra, 2 * s + hh));
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(B + off(rb, 2 * s + hh));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
  }
  // acc[h][col] += sum_k IMG[jrow0 + k][hsub*32 + h] * X[k][col], k = 0..31, X = accumulator tile in registers.
  // Per k-step s and half t, every 16-lane group reads one 4-row x 16-column block transposed:
  // lane 4q+p of the group supplies the address of (row q, columns 4p..4p+3); lane i receives column i.
  __device__ static __forceinline__ void tile_xb_tr(f32x16& acc, const __bf16* img, int hsub, int jrow0,
                                                    const f32x16& x) {
    const int l = xf_lane(), g16 = l >> 4, hh = g16 >> 1, li = l & 15, q = li >> 2, p = li & 3;
    const int col = hsub * 32 + 16 * (g16 & 1) + 4 * p;
    const int c = col >> 3, within = col & 7;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 b;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (__bf16)x[8 * s + j];
      union { xf_s16x4 v[2]; bf16x8 f; } a;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = jrow0 + 16 * s + 8 * t + 4 * hh + q;
        const __bf16* ptr = img + off(row, c) + within;
        a.v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) xf_s16x4*)(ptr));
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, b, acc, 0, 0, 0);
    }
  }
  // One wave instruction of the gather: lane -> (row = row0 + lane / CPR, position = lane % CPR);
  // returns the SOURCE chunk index this lane must fetch so that position holds chunk (position ^ swizzle).
  __device__ static __forceinline__ int gather_row(int row0) { return row0 + xf_lane() / CPR; }
  __device__ static __forceinline__ int gather_src_chunk(int row) { return (xf_lane() % CPR) ^ swz(row); }
  static constexpr int kRowsPerInstr = 64 / CPR;            // rows covered by one 1-KiB wave instruction
};

// 16 bytes per lane global -> LDS without registers; lds_base must be wave-uniform.
__device__ __forceinline__ void xf_glds16(const void* gsrc, void* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// The same instruction issued from an asm statement, i.e. outside hipcc's s_waitcnt bookkeeping: while a
// builtin-issued LDS-DMA is in flight hipcc puts s_waitcnt vmcnt(0) in front of LDS reads it cannot prove disjoint
// from the DMA target (here: every read of per-negative side data, twice per 32 x 32 sub-block), which drains the
// prefetch. The caller owns the ordering: s_waitcnt vmcnt(N) + barrier before anyone reads the destination.
// M0 (the LDS destination base) is compiler-reserved: it is saved, set and restored inside the one statement.
// ... with the address as a wave-uniform 64-bit base (SGPR pair) + a 32-bit unsigned byte offset per lane: half the
// address registers of the 64-bit per-lane form and no 64-bit vector arithmetic per piece (the source must lie within
// 4 GiB of the base).
__device__ __forceinline__ void xf_glds16_raw_so(const void* sbase, uint32_t voff, void* lds_base) {
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_base;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(dst)
               : "memory");
}
__device__ __forceinline__ void xf_glds16_raw(const void* gsrc, void* lds_base) {
  const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds_base;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
