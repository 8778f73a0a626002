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
  using elem = typename P::elem;
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

}  // namespace

extern "C" int xfmr_selftest_mfma(int32_t* out, void* stream) {
  if (!out) return XFMR_EINVAL;
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
