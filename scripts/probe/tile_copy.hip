// Micro-benchmark: copy an M x PITCH-byte matrix tile by tile (TR rows x TC bytes per 256-thread workgroup), in the
// GEMM epilogue's access pattern, to see what the partial-row pattern costs against a contiguous stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void tile_copy(const uint4* src, uint4* dst, int M, int pitch16, int tr, int tc16,
                                                 int nt_n, int nt_m, int xcd_order) {
  int d = blockIdx.x, n, m;
  if (xcd_order) { int xcd = d & 7, slot = d >> 3; n = slot % nt_n; m = (slot / nt_n) * 8 + xcd; }
  else { n = d % nt_n; m = d / nt_n; }
  if (m >= nt_m) return;
  const int per = tr * tc16;
  for (int i = threadIdx.x; i < per; i += 256) {
    const int r = i / tc16, c = i % tc16;
    const long row = (long)m * tr + r;
    if (row < M) {
      const long o = row * pitch16 + (long)n * tc16 + c;
      dst[o] = src[o];
    }
  }
}
int main() {
  const int M = 102400;
  for (int pitch : {256, 1024}) {
    const size_t bytes = (size_t)M * pitch;
    uint4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    for (int tc : {128, 256, 512, 1024}) {
      if (tc > pitch) continue;
      for (int tr : {16, 64}) {
        for (int xo : {0, 1}) {
          const int nt_n = pitch / tc, nt_m = (M + tr - 1) / tr;
          const int grid = xo ? ((nt_m + 7) / 8) * 8 * nt_n : nt_n * nt_m;
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          std::vector<float> ts;
          for (int it = 0; it < 12; ++it) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(tile_copy, dim3(grid), dim3(256), 0, 0, a, b, M, pitch / 16, tr, tc / 16, nt_n, nt_m, xo);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2) ts.push_back(ms);
          }
          std::sort(ts.begin(), ts.end());
          const float ms = ts[ts.size() / 2];
          printf("pitch %4d B  tile %3d rows x %4d B  xcd_order %d : %7.1f us  %.2f TB/s (read+write)\n", pitch, tr, tc, xo,
                 ms * 1e3, 2.0 * bytes / (ms * 1e-3) / 1e12);
        }
      }
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
