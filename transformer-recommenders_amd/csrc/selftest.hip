// On-device self-test of the MFMA operand / accumulator lane maps the library is built on
// (PrecBF16 / PrecF32 :: tile_nt, tile_nreg, tile_xb and xf_acc_row). Small exact integers, asymmetric
// operands, compared against a scalar evaluation: a swapped row/col or a wrong k-permutation cannot pass.
#include "common.h"

namespace {

__device__ __forceinline__ float tval(int which, int r, int c) {
  // asymmetric small integers in [-4, 4]; exactly representable in bf16, sums stay exact in fp32
  uint32_t h = xf_hash32((uint32_t)(which * 7919 + r * 131 + c * 17 + 3));
  return (float)((int)(h % 9u) - 4);
}

template <class P>
__device__ int run_checks(typename P::elem* sA, typename P::elem* sB, float* sRow) {
  constexpr int K = 32, LD = xf_ld<P>(K), LDTT = 32 + 4;
  const int lane = xf_lane();
  int bad = 0;
  // images: A[32][K], B[32][K]
  for (int i = lane; i < 32 * K; i += 64) {
    const int r = i / K, c = i % K;
    sA[r * LD + c] = P::cvt(tval(1, r, c));
    sB[r * LD + c] = P::cvt(tval(2, r, c));
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // (1) tile_nt: C[i][j] = sum_k A[i][k] B[j][k]
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  P::tile_nt(acc, sA, LD, 0, sB, LD, 0, K);
  for (int r = 0; r < 16; ++r) {
    const int i = xf_acc_row(r, lane), j = lane & 31;
    float ref = 0.f;
    for (int k = 0; k < K; ++k) ref += tval(1, i, k) * tval(2, j, k);
    bad += (acc[r] != ref);
  }
  // (2) tile_nreg: B rows taken from registers loaded by RegRows from a global-like fp32 row
  for (int i = lane; i < 32 * K; i += 64) sRow[i] = tval(2, i / K, i % K);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  RegRows<P, K> breg;
  breg.load(sRow + (lane & 31) * K, true);
  f32x16 acc2;
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  P::tile_nreg(acc2, sA, LD, 0, breg.regs(), K);
  for (int r = 0; r < 16; ++r) bad += (acc2[r] != acc[r]);
  // (3) tile_xb: Y[i][col] = sum_k A'[i][k] X[k][col], X = acc (32x32 in accumulator layout), A' image [32][32]
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < 32 * 32; i += 64) sB[(i / 32) * LDTT + (i % 32)] = P::cvt(tval(3, i / 32, i % 32));
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  f32x16 x;
  for (int r = 0; r < 16; ++r) x[r] = tval(4, xf_acc_row(r, lane), lane & 31);
  f32x16 y;
  for (int r = 0; r < 16; ++r) y[r] = 0.f;
  P::tile_xb(y, sB, LDTT, 0, 0, x);
  for (int r = 0; r < 16; ++r) {
    const int i = xf_acc_row(r, lane), col = lane & 31;
    float ref = 0.f;
    for (int k = 0; k < 32; ++k) ref += tval(3, i, k) * tval(4, k, col);
    bad += (y[r] != ref);
  }
  return bad;
}

__global__ __launch_bounds__(64) void selftest_kernel(int* out) {
  __shared__ __attribute__((aligned(16))) float sA[32 * 40];
  __shared__ __attribute__((aligned(16))) float sB[32 * 40];
  __shared__ __attribute__((aligned(16))) float sRow[32 * 32];
  int bad_bf16 = run_checks<PrecBF16>(reinterpret_cast<__bf16*>(sA), reinterpret_cast<__bf16*>(sB), sRow);
  __builtin_amdgcn_wave_barrier();
  int bad_f32 = run_checks<PrecF32>(sA, sB, sRow);
  for (int o = 32; o > 0; o >>= 1) {
    bad_bf16 += __shfl_xor(bad_bf16, o, 64);
    bad_f32 += __shfl_xor(bad_f32, o, 64);
  }
  if (threadIdx.x == 0) {
    out[0] = bad_bf16 + bad_f32;
    out[1] = bad_bf16;
    out[2] = bad_f32;
  }
}

// ---- swizzled LDS-DMA image: gather layout, row-operand reads, ds_read_b64_tr_b16 transposed reads ----------
constexpr int SH = 128, SROWS = 64;
__global__ void selftest_fill_kernel(__bf16* src) {
  for (int i = threadIdx.x; i < SROWS * SH; i += blockDim.x) src[i] = (__bf16)tval(5, i / SH, i % SH);
}
__global__ __launch_bounds__(64) void selftest_swz_kernel(const __bf16* src, int* out) {
  using SI = SwzImg<SH>;
  __shared__ __attribute__((aligned(16))) __bf16 img[SROWS * SH];
  __shared__ __attribute__((aligned(16))) float sRow[32 * SH];
  const int lane = xf_lane();
  int bad = 0;
  // gather rows in a permuted order (row r of the image <- source row perm(r)) through LDS-DMA
  for (int q = 0; q < SROWS / SI::kRowsPerInstr; ++q) {
    const int r = SI::gather_row(q * SI::kRowsPerInstr);
    const int srow = (r * 7 + 3) % SROWS;
    xf_glds16(src + srow * SH + 8 * SI::gather_src_chunk(r), img + q * 512);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // (a) placement: logical element (r, col) sits at off(r, col/8) + col%8
  for (int i = lane; i < SROWS * SH; i += 64) {
    const int r = i / SH, col = i % SH;
    bad += ((float)img[SI::off(r, col >> 3) + (col & 7)] != tval(5, (r * 7 + 3) % SROWS, col));
  }
  // (b) row-operand product: C[j][i] = sum_h IMG[32 + j][h] * B[i][h], B rows via RegRows
  for (int i = lane; i < 32 * SH; i += 64) sRow[i] = tval(2, i / SH, i % SH);
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  RegRows<PrecBF16, SH> breg;
  breg.load(sRow + (lane & 31) * SH, true);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  SI::tile_nreg(acc, img, 32, breg.regs());
  for (int r = 0; r < 16; ++r) {
    const int j = 32 + xf_acc_row(r, lane), i = lane & 31;
    float ref = 0.f;
    for (int h = 0; h < SH; ++h) ref += tval(5, (j * 7 + 3) % SROWS, h) * tval(2, i, h);
    bad += (acc[r] != ref);
  }
  // (c) transposed product: Y[h][col] = sum_k IMG[32 + k][hsub*32 + h] * X[k][col]
  f32x16 x;
  for (int r = 0; r < 16; ++r) x[r] = tval(4, xf_acc_row(r, lane), lane & 31);
  for (int hsub = 0; hsub < SH / 32; ++hsub) {
    f32x16 y;
    for (int r = 0; r < 16; ++r) y[r] = 0.f;
    SI::tile_xb_tr(y, img, hsub, 32, x);
    for (int r = 0; r < 16; ++r) {
      const int h = hsub * 32 + xf_acc_row(r, lane), col = lane & 31;
      float ref = 0.f;
      for (int kk = 0; kk < 32; ++kk) ref += tval(5, ((32 + kk) * 7 + 3) % SROWS, h) * tval(4, kk, col);
      bad += (y[r] != ref);
    }
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if (lane == 0) out[3] = bad;
}

}  // namespace

extern "C" int xfmr_selftest_mfma(int32_t* out, void* stream) {
  if (!out) return XFMR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, out);
  XF_LAUNCH_CHECK();
  __bf16* src = reinterpret_cast<__bf16*>(out + 4);  // scratch: 64 x 128 bf16 = 16 KiB behind the 4 result words
  hipLaunchKernelGGL(selftest_fill_kernel, dim3(1), dim3(256), 0, st, src);
  XF_LAUNCH_CHECK();
  hipLaunchKernelGGL(selftest_swz_kernel, dim3(1), dim3(64), 0, st, (const __bf16*)src, out);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
