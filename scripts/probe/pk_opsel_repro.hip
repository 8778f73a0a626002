// Standalone re-qualification probe for the packed-fp32 op_sel form that scripts/check_isa.py keeps out of the library
// (every build of the LayerNorm-backward GEMM epilogue that contained
//   v_pk_add_f32 vD, vD, vB op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]   (x - mean: low result lane reads the HIGH dword of vB)
//   v_pk_mul_f32 vE, vC, vD op_sel:[1,0]                             (* rstd: low result lane reads the HIGH dword of vC)
// gave 1-5 wrong rows of 102 400 per launch, with the consumers 2-4 issue slots behind the producer).
//
// ONE launch of a large grid: every lane evaluates, on hashed inputs,
//   bad form    the two packed instructions above, the consumer DIST independent vector instructions behind the producer
//   pinned form the same arithmetic with op_sel_hi broadcasts of 32-bit registers (what XF_PIN_SCALAR makes hipcc emit)
//   scalar form four v_add_f32 / v_mul_f32
// and counts, per DIST in 0..4 and per context (plain VALU stream / behind an LDS crossbar op / behind an MFMA), the
// lanes whose bad-form or pinned-form bits differ from the scalar form. Exit status 0 = no difference anywhere.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/probe/pk_opsel_repro.hip -o build/pk_opsel_repro && build/pk_opsel_repro
// Prints the hipcc / ROCm versions it was built with beside the counts, so a toolchain bump is one command to re-check.
#include <hip/hip_runtime.h>
#include <hip/hip_version.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float unit(uint32_t h) { return (float)(h >> 8) * (1.f / 8388608.f) - 1.f; }  // [-1, 1)

// DIST independent vector instructions between producer and consumer (they touch only the filler registers)
#define FILL0 ""
#define FILL1 "v_add_f32 %[f0], %[f0], %[f1]\n"
#define FILL2 FILL1 "v_mul_f32 %[f1], %[f1], %[f0]\n"
#define FILL3 FILL2 "v_add_f32 %[f0], %[f0], %[f1]\n"
#define FILL4 FILL3 "v_mul_f32 %[f1], %[f1], %[f0]\n"

#define BAD_FORM(FILL)                                                                                      \
  asm volatile("v_pk_add_f32 %[d], %[d], %[b] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n" FILL \
               "v_pk_mul_f32 %[e], %[c], %[d] op_sel:[1,0] op_sel_hi:[1,1]\n"                               \
               : [d] "+v"(d), [e] "=&v"(e), [f0] "+v"(f0), [f1] "+v"(f1)                                    \
               : [b] "v"(b), [c] "v"(c))
// pinned: the broadcast operand is a 32-bit register of its own, read by BOTH result lanes through op_sel_hi:[.,0]
#define PINNED_FORM(FILL)                                                                                   \
  asm volatile("v_pk_add_f32 %[d], %[d], %[bh] op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n" FILL \
               "v_pk_mul_f32 %[e], %[ch], %[d] op_sel:[0,0] op_sel_hi:[0,1]\n"                              \
               : [d] "+v"(d), [e] "=&v"(e), [f0] "+v"(f0), [f1] "+v"(f1)                                    \
               : [bh] "v"(bh2), [ch] "v"(ch2))

template <int CTX>
__device__ __forceinline__ void context(float& f0, float& f1, int lane) {
  if (CTX == 1) {  // an LDS-crossbar instruction in flight (the epilogue's row sums went through ds_bpermute / DPP)
    f0 += __shfl_xor(f1, 16, 64);
  } else if (CTX == 2) {  // an MFMA in flight (the epilogue follows the tile's last matrix instructions)
    bf16x8 a, bb;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)f0; bb[i] = (__bf16)f1; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
    f1 += acc[lane & 15] * 1e-30f;
  }
}

template <int CTX>
__global__ __launch_bounds__(256) void probe(unsigned long long* counts, int iters) {
  const int lane = threadIdx.x & 63;
  uint32_t key = hash32(blockIdx.x * 256u + threadIdx.x + 0x9e3779b9u * (CTX + 1));
  unsigned bad[5] = {0, 0, 0, 0, 0}, pin[5] = {0, 0, 0, 0, 0};
  float f0 = unit(key ^ 1), f1 = unit(key ^ 2);
  for (int it = 0; it < iters; ++it) {
    key = hash32(key + it);
    const f32x2 d0 = {unit(hash32(key ^ 11)), unit(hash32(key ^ 12))};
    const f32x2 b = {unit(hash32(key ^ 13)), unit(hash32(key ^ 14))};
    const f32x2 c = {unit(hash32(key ^ 15)), unit(hash32(key ^ 16))};
    const f32x2 bh2 = {b[1], b[1]}, ch2 = {c[1], c[1]};
    // scalar form
    const float r0 = d0[0] - b[1], r1 = d0[1] - b[1];
    const float s0 = c[1] * r0, s1 = c[1] * r1;
#define ONE(DIST, FILL)                                                                     \
    {                                                                                       \
      f32x2 d = d0, e;                                                                      \
      context<CTX>(f0, f1, lane);                                                           \
      BAD_FORM(FILL);                                                                       \
      bad[DIST] += (__float_as_uint(e[0]) != __float_as_uint(s0)) | (__float_as_uint(e[1]) != __float_as_uint(s1)) | \
                   (__float_as_uint(d[0]) != __float_as_uint(r0)) | (__float_as_uint(d[1]) != __float_as_uint(r1));  \
      d = d0;                                                                               \
      context<CTX>(f0, f1, lane);                                                           \
      PINNED_FORM(FILL);                                                                    \
      pin[DIST] += (__float_as_uint(e[0]) != __float_as_uint(s0)) | (__float_as_uint(e[1]) != __float_as_uint(s1)) | \
                   (__float_as_uint(d[0]) != __float_as_uint(r0)) | (__float_as_uint(d[1]) != __float_as_uint(r1));  \
    }
    ONE(0, FILL0) ONE(1, FILL1) ONE(2, FILL2) ONE(3, FILL3) ONE(4, FILL4)
#undef ONE
  }
  if (f0 == 12345.f && f1 == 54321.f) bad[0] += 1;  // keep the filler chain alive
  for (int k = 0; k < 5; ++k) {
    if (bad[k]) atomicAdd(&counts[CTX * 10 + k], (unsigned long long)bad[k]);
    if (pin[k]) atomicAdd(&counts[CTX * 10 + 5 + k], (unsigned long long)pin[k]);
  }
}

int main() {
  unsigned long long* counts;
  if (hipMalloc(&counts, 30 * sizeof(*counts)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
  hipMemset(counts, 0, 30 * sizeof(*counts));
  const int blocks = 65536, iters = 64;  // 65536 x 256 lanes x 64 iterations x 5 distances per context, ONE launch each
  hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, counts, iters);
  hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, counts, iters);
  hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, counts, iters);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  unsigned long long h[30];
  hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int rt = 0;
  hipRuntimeGetVersion(&rt);
  printf("pk_opsel_repro: device %s (%s), HIP build %d.%d.%d, runtime %d, clang %s\n", prop.name, prop.gcnArchName, HIP_VERSION_MAJOR,
         HIP_VERSION_MINOR, HIP_VERSION_PATCH, rt, __clang_version__);
  const char* ctx[3] = {"plain VALU stream", "behind ds_bpermute", "behind an MFMA"};
  unsigned long long bad_total = 0, pin_total = 0;
  const double evals = (double)blocks * 256 * iters;
  for (int c = 0; c < 3; ++c) {
    printf("  %-20s consumer distance 0..4: bad form mismatches", ctx[c]);
    for (int k = 0; k < 5; ++k) { printf(" %llu", h[c * 10 + k]); bad_total += h[c * 10 + k]; }
    printf(" | pinned form");
    for (int k = 0; k < 5; ++k) { printf(" %llu", h[c * 10 + 5 + k]); pin_total += h[c * 10 + 5 + k]; }
    printf("  (of %.3g evaluations each)\n", evals);
  }
  printf("RESULT bad_form_mismatches=%llu pinned_form_mismatches=%llu\n", bad_total, pin_total);
  return (bad_total || pin_total) ? 1 : 0;
}
