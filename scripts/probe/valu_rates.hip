// Issue cost of the vector instructions the loss epilogues are made of, on gfx950 (MI355X): cycles one SIMD is held
// per wave64 instruction, measured as wall time of a kernel that issues nothing else (independent chains, every SIMD
// of the chip loaded with WAVES waves).   hipcc --offload-arch=gfx950 -O3 -o build/valu_rates scripts/probe/valu_rates.hip
// Printed: ns per wave-instruction per SIMD and the same relative to v_fma_f32. (The guide quotes 2 cycles per wave64
// op; the logging pass's counters say 4.35 on its mix: this probe settles what each instruction costs.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 64;  // instructions per loop trip (8 chains x 8)

#define BODY8(INS)  INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int trips, float seed) {
  float a[8]; f2 p[8];
  const float x = seed + threadIdx.x * 1e-3f, y = 1.0001f;
  f2 x2 = {x, x + 1.f}, y2 = {y, y};
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = x + i; p[i] = f2{x + i, x - i}; }
  unsigned long long m = 0;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int r = 0; r < UNROLL / 8; ++r) {
      if (OP == 0) {
#define I0(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        BODY8(I0)
      } else if (OP == 1) {
#define I1(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(y2));
        BODY8(I1)
      } else if (OP == 2) {
#define I2(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        BODY8(I2)
      } else if (OP == 3) {
#define I3(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
        BODY8(I3)
      } else if (OP == 4) {
#define I4(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : );
        BODY8(I4)
      } else if (OP == 5) {
#define I5(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
        BODY8(I5)
      } else if (OP == 6) {
#define I6(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
        BODY8(I6)
      } else if (OP == 7) {
#define I7(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
        BODY8(I7)
      } else if (OP == 8) {
#define I8(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        BODY8(I8)
      } else if (OP == 9) {
#define I9(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(x2), "v"(y2));
        BODY8(I9)
      } else if (OP == 10) {
#define I10(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        BODY8(I10)
      } else if (OP == 11) {
#define I11(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        BODY8(I11)
      } else if (OP == 12) {
#define I12(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        BODY8(I12)
      } else if (OP == 13) {
#define I13(i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
        BODY8(I13)
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s + (float)m;  // keep the chains alive
}

template <int OP>
double run(const char* name, int waves_per_simd, double base) {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount, simds = cus * 4;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
  const int trips = 4000;
  float* out; CK(hipMalloc(&out, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<OP><<<blocks, 256>>>(out, 200, 1.f);
  CK(hipDeviceSynchronize());
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(out, trips, 1.f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double wave_instr_per_simd = (double)waves_per_simd * trips * UNROLL;
  const double ns = best * 1e6 / wave_instr_per_simd;
  printf("%-34s waves/SIMD %d: %.3f ns per wave-instruction per SIMD = %.2f cycles at %.2f GHz%s", name, waves_per_simd, ns,
         ns * pr.clockRate * 1e-6, pr.clockRate * 1e-6, base > 0 ? "" : "\n");
  if (base > 0) printf("  (%.2f x v_fma_f32)\n", ns / base);
  CK(hipFree(out));
  return ns;
}

int main() {
  for (int w : {1, 2, 4}) {
    const double b = run<0>("v_fma_f32", w, 0);
    run<1>("v_pk_fma_f32", w, b);
    run<9>("v_pk_fma_f32 op_sel_hi:[1,0,1]", w, b);
    run<6>("v_pk_mul_f32", w, b);
    run<7>("v_pk_add_f32", w, b);
    run<8>("v_max_f32", w, b);
    run<11>("v_max3_f32", w, b);
    run<4>("v_cndmask_b32 (vcc)", w, b);
    run<5>("v_cmp_lt_f32 -> vcc", w, b);
    run<2>("v_exp_f32", w, b);
    run<3>("v_log_f32", w, b);
    run<10>("v_rcp_f32", w, b);
    run<13>("v_cvt_pk_bf16_f32", w, b);
  }
  return 0;
}
