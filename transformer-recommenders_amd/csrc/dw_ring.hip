// Weight-gradient GEMMs of the encoder backward, round 4 form: dW[N][K] = dy^T[N][T] x[T][K] in split-T slabs, the operands
// streamed through a four-stage LDS-DMA ring (xfmr_encoder_bwd; reference: torch.autograd through nn.Linear,
// models.py:93-102 / modeling_bert.py:111-448 -- the grad_weight = grad_output^T @ input of every Linear).
//
// Why a second kernel beside gemm.hip's generic one (gemm_kernel<..., EPI_SPLITK>): there each 128-token slice goes
// global -> registers -> LDS with two barriers and every piece's address arithmetic in the loop -- 12.6 vector instructions
// per MFMA, the waves waiting 41 % of their resident cycles (profiles/r02_step_pmc.md), 117 us per layer at the benchmark
// shape for 419 MB of operands = 3.5 TB/s. Here
//   * a workgroup of EIGHT waves owns a 128 x 128 tile of dW and one token slab [t0, t1) of the split plan (the same plan:
//     xf_dw_split_plan, same slab buffers), and walks it in stages of 64 tokens;
//   * a stage is the [64 tokens][128 features] pieces of dy and of x, fetched by global_load_lds_dwordx4 (no registers,
//     4 wave instructions per wave and stage) into a ring of four stage buffers: three stages are in flight while one
//     is multiplied -- one barrier per stage, s_waitcnt vmcnt(8 / 4 / 0) by hand (the asm-issued DMA is outside hipcc's
//     own bookkeeping, common.h);
//   * both MFMA operands contract over the image's ROW index (tokens), so both come from ds_read_b64_tr_b16 on the swizzled
//     images (SwzImg<128>: the gather applies the XOR on the source side) -- 6 transposed reads per two MFMAs;
//   * the Linear's bias gradient (column sums of dy over the slab) is two more MFMAs per step against a register of ones
//     in the waves of the tile's first column block, instead of a bf16 -> fp32 conversion and an add per element.
// Per CU one workgroup (128 KB of LDS), two waves per SIMD. MEASURED at the benchmark shape (T = 102 400, one layer's four
// weights, scripts/probe/dw_ring_probe.py): 22 / 13.5 / 27 / 25.5 us per weight launched alone (QKV, out-proj, FFN1, FFN2:
// 4.7-5.1 TB/s of operand bytes) against 29 / 14.5 / 36 / 30.5 for the generic kernel (3.6-4.3); as ONE grouped launch 101-105
// against 114-118 us. Step: 3.20 against 3.27 ms (batch 512), 1.97 against 2.00 on MovieLens-like packed batches.
// Slabs: the same [splits][N][K] fp32 partial products (and [splits][N] bias rows) the generic kernel writes, reduced by
// the backward's one xf_multi_rowsum launch -- deterministic, bit-reproducible run to run. Differs from the generic kernel
// in the last bits only (the 16 tokens of an MFMA step sit in another order in the operand registers).
#include "internal.h"

namespace {

// Stage depth and ring length are compile-time switches for A/B builds (scripts/build_variant.sh). MEASURED (round 4, one
// box, batch 512, alternating runs; ms per step against the generic kernel's 3.26-3.28): 64 tokens x 4 stages (128 KB, one
// workgroup per CU) 3.20-3.21; 64 x 2 (64 KB, two per CU) 3.25-3.29; 32 x 4 (64 KB, two per CU) 3.24-3.28. In isolation a
// weight takes the same time with 2, 3 or 4 stages of 64 tokens, and the same with three quarters of the operand reads
// compiled out: at 4.7-5.1 TB/s of operands (7 TB/s out of L2 with the tiles' re-reads) the launch sits at the memory
// system, not at the ring depth or the LDS port. (Also measured: even / odd 16-token steps on separate accumulators -- four
// independent MFMA chains per wave instead of two -- 36.7 against 36.7 us on a 102-workgroup launch: the SQ counters' 45 %
// "issue stalled" of this kernel, profiles/r04_step_pmc.md, is not the accumulate chain.)
#ifndef XF_DWR_RT
#define XF_DWR_RT 64
#endif
constexpr int RT = XF_DWR_RT;  // tokens per stage
#ifndef XF_DWR_NST
#define XF_DWR_NST 4
#endif
constexpr int NST = XF_DWR_NST;  // ring stages
constexpr int IPW = RT / 16;   // gather instructions per wave and stage
constexpr int TW = 128;   // tile width in features, both ways
using Img = SwzImg<TW>;
constexpr int IMG_ELEMS = RT * TW;  // one stage image: [RT][128] bf16

struct DwRingItem {
  const __bf16* dy;  // [T][N]
  const __bf16* x;   // [T][K]
  float* slabs;      // [splits][N][K]
  float* bias_part;  // [splits][N] or null
  int N, K, k_chunk, splits;
};
struct DwRingArgs {
  DwRingItem it[4];
  int start[5];  // workgroups [start[i], start[i + 1]) are item i's
  int64_t T;
};

// MFMA operand of one 32-feature block for the 16 tokens from `jrow`: image[jrow + k][hsub * 32 + h], h on the lane,
// k in the register (SwzImg::tile_xb_tr's read; the token order inside the register is the same for both operands)
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, const int hsub, const int jrow) {
  const int l = xf_lane(), g16 = l >> 4, hh = g16 >> 1, li = l & 15, q = li >> 2, p = li & 3;
  const int col = hsub * 32 + 16 * (g16 & 1) + 4 * p;
  const int c = col >> 3, within = col & 7;
  union { xf_s16x4 v[2]; bf16x8 f; } a;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = jrow + 8 * t + 4 * hh + q;
    a.v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) xf_s16x4*)(img + Img::off(row, c) + within));
  }
  return a.f;
}

template <int N>
__device__ __forceinline__ void wait_vm() {  // s_waitcnt vmcnt(N) only (gfx9 encoding: expcnt / lgkmcnt fields all ones)
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (7 << 4) | (0xF << 8));
}

__global__ __launch_bounds__(512, (RT * NST <= 128) ? 4 : 2) void dw_ring_kernel(const DwRingArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* const sA = reinterpret_cast<__bf16*>(smem_raw);        // [NST][64][128] dy pieces
  __bf16* const sB = sA + NST * IMG_ELEMS;                       // [NST][64][128] x pieces
  const int b = (int)blockIdx.x;
  const int ii = (b >= p.start[1]) + (b >= p.start[2]) + (b >= p.start[3]);
  const DwRingItem it = p.it[ii];
  const int d = b - p.start[ii], xcd = d & 7, slot = d >> 3;
  const int nt_n = it.N / TW, nt_k = it.K / TW, per = nt_n * nt_k;
  const int z = (slot / per) * 8 + xcd;  // the tiles of one slab on one XCD, back to back: they share its operand rows in L2
  if (z >= it.splits) return;            // (whole workgroup)
  const int rr = slot % per, tn = rr / nt_k, tk = rr % nt_k;
  const int64_t t0 = (int64_t)z * it.k_chunk;
  const int64_t t1 = (t0 + it.k_chunk < p.T) ? t0 + it.k_chunk : p.T;
  const int nst = (int)((t1 - t0 + RT - 1) / RT);

  const int lane = xf_lane(), wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wid >> 2, wc = wid & 3;  // wave tile: dW rows [64 wr, +64) x columns [32 wc, +32) of the 128 x 128 tile

  // ---- the gather: waves 0-3 fetch the dy image, waves 4-7 the x image; four 1-KiB instructions each per stage (4 token
  // rows of 256 B per instruction: lane -> row lane / 16, 16-byte position lane % 16, source chunk = position ^ swizzle)
  const bool isB = wid >= 4;
  const __bf16* const gbase = isB ? it.x : it.dy;
  const int ld = isB ? it.K : it.N, col0 = (isB ? tk : tn) * TW;
  __bf16* const simg = isB ? sB : sA;
  const int grow = 4 * IPW * (wid & 3) + (lane >> 4);  // stage row of this lane's piece in instruction 0 (+4 per instruction)
  uint32_t cofs[IPW];                                  // byte offset of the piece inside its token row
#pragma unroll
  for (int j = 0; j < IPW; ++j) cofs[j] = (uint32_t)(col0 + 8 * ((lane & 15) ^ Img::swz(grow + 4 * j))) * 2u;
  auto issue = [&](const int s) {  // stage s -> ring buffer s % NST
    __bf16* const dst = simg + (s % NST) * IMG_ELEMS + (4 * IPW * (wid & 3)) * TW;
    const int64_t ts = t0 + (int64_t)s * RT;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      int64_t tok = ts + grow + 4 * j;
      tok = tok < t1 ? tok : t1 - 1;  // rows past the slab: a valid address, zeroed after they land
      xf_glds16_raw_so(gbase, (uint32_t)(tok * ld * 2) + cofs[j], dst + 4 * j * TW);
    }
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nst) issue(s);

  f32x16 acc[2], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accb[i][r] = 0.f; }
  // the bias gradient's extra MFMAs are dealt over the four waves that hold the same dy rows (wave wc takes every fourth
  // 16-token step); their partial sums meet in LDS after the loop. (All on the wc == 0 waves -- waves 0 and 4, the same
  // SIMD -- doubled that SIMD's matrix work.)
  const bool do_bias = it.bias_part != nullptr && tk == 0;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  for (int s = 0; s < nst; ++s) {
    // this wave's pieces of stage s have landed once at most the later stages' are outstanding
    const int later = (nst - 1 - s) < (NST - 2) ? (nst - 1 - s) : (NST - 2);
    if (NST >= 4 && later >= 2) wait_vm<2 * IPW>();
    else if (NST >= 3 && later >= 1) wait_vm<IPW>();
    else wait_vm<0>();
    __syncthreads();  // everyone's pieces of stage s are in LDS; everyone is done reading stage s - 1's buffer
    const __bf16* const a = sA + (s % NST) * IMG_ELEMS;
    const __bf16* const bm = sB + (s % NST) * IMG_ELEMS;
    if (s + NST - 1 < nst) issue(s + NST - 1);  // into the buffer stage s - 1 used
    const int valid = (int)(t1 - (t0 + (int64_t)s * RT));  // token rows of this stage inside the slab
    if (valid < RT) {  // (the slab's last stage only) rows past the end hold a copy of the last row: zero them
      for (int c = (int)threadIdx.x; c < (RT - valid) * 32; c += 512) {
        const int row = valid + (c >> 5), piece = c & 31;  // 32 16-byte pieces per row: 16 of the dy image, 16 of x
        __bf16* img = (piece < 16 ? sA : sB) + (s % NST) * IMG_ELEMS;
        *reinterpret_cast<uint4*>(img + row * TW + 8 * (piece & 15)) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    bf16x8 fa[2][2], fb[2];
    fa[0][0] = frag_tr(a, 2 * wr, 0);
    fa[0][1] = frag_tr(a, 2 * wr + 1, 0);
    fb[0] = frag_tr(bm, wc, 0);
#pragma unroll
    for (int ks = 0; ks < RT / 16; ++ks) {
      if (ks + 1 < RT / 16) {  // the next step's operands are read while this step's MFMAs run
        fa[(ks + 1) & 1][0] = frag_tr(a, 2 * wr, 16 * (ks + 1));
        fa[(ks + 1) & 1][1] = frag_tr(a, 2 * wr + 1, 16 * (ks + 1));
        fb[(ks + 1) & 1] = frag_tr(bm, wc, 16 * (ks + 1));
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][0], fb[ks & 1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][1], fb[ks & 1], acc[1], 0, 0, 0);
      if (do_bias && ((s * (RT / 16) + ks) & 3) == wc) {  // (wave-uniform) row sums of dy^T: every column of the product with ones
        accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][0], ones, accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][1], ones, accb[1], 0, 0, 0);
      }
    }
  }

  // ---- the slab tile, straight from the accumulator layout: register r of lane l = row (r&3) + 8 (r>>2) + 4 (l>>5),
  // column l & 31 -- 32 lanes write 128 contiguous bytes of a row
  float* const slab = it.slabs + (int64_t)z * it.N * it.K;
  const int kcol = tk * TW + 32 * wc + (lane & 31);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n0 = tn * TW + 64 * wr + 32 * i + 4 * (lane >> 5);
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(int64_t)(n0 + (r & 3) + 8 * (r >> 2)) * it.K + kcol] = acc[i][r];
  }
  if (do_bias) {  // (workgroup-uniform) the four waves' shares of the 128 row sums, added in a fixed order
    __syncthreads();  // everyone is done with the stage images
    float* const red = reinterpret_cast<float*>(smem_raw);  // [4 wc][128 rows]
    if ((lane & 31) == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wc * TW + 64 * wr + 32 * i + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2)] = accb[i][r];
    }
    __syncthreads();
    if (threadIdx.x < TW) {
      const int n = (int)threadIdx.x;
      it.bias_part[(int64_t)z * it.N + tn * TW + n] = (red[n] + red[TW + n]) + (red[2 * TW + n] + red[3 * TW + n]);
    }
  }
}

}  // namespace

extern "C" {

// Whether the ring kernel takes these weight-gradient GEMMs (bf16 operands in HBM, 128-wide tiles both ways, operands
// within the 32-bit byte offsets of the gather); XFMR_DW_RING=0 keeps the generic kernel (A/B runs).
bool xf_dw_ring_takes(const XfDwItem* items, int n, int64_t M, int32_t precision, uint32_t s16) {
  static const bool on = [] { const char* e = getenv("XFMR_DW_RING"); return !(e && *e == '0'); }();
  constexpr uint32_t SAB = XF_S16_A | XF_S16_B;
  if (!on || !items || n < 1 || n > 4 || precision != XFMR_PREC_BF16 || (s16 & SAB) != SAB || M <= 0) return false;
  for (int i = 0; i < n; ++i) {
    const XfDwItem& t = items[i];
    if (!t.dy || !t.x || !t.slabs || !t.splits || t.N <= 0 || t.K <= 0 || (t.N % TW) || (t.K % TW)) return false;
    if (!xf_aligned16(t.dy) || !xf_aligned16(t.x) || !xf_aligned16(t.slabs)) return false;
    const int64_t widest = t.N > t.K ? t.N : t.K;
    if ((uint64_t)M * (uint64_t)widest * 2u >= (1ull << 32)) return false;
  }
  return true;
}

// The ring kernel's own slab plan. The generic plan (xf_dw_split_plan: ~1024 workgroups of 128 x 64 tiles, three per CU) cuts
// the tokens into 62-115 slabs per weight at the benchmark shape and 25-50 at small batches; this kernel runs ONE workgroup
// per CU on 128 x 128 tiles, and a slab costs it a ring fill, a 64 KB tile store and later a row of the reduction launch:
// ~3 000 tokens per slab, but at least 64 workgroups per weight (128 x 128 tiles x slabs), never more slabs than the generic plan
// (the slab room is carved for that: xfmr_linear_bwd_dw_workspace). MEASURED (round 4, scripts/probe/dwr_plan_sweep.sh, one
// box, two rounds, ms per step against the generic plan): batch 128 1.166-1.170 against 1.210; batch 32 0.656-0.670 against
// 0.690-0.692; MovieLens-like packed batches of 512 (~50 k tokens) 1.94-1.95 against 1.96-1.97; batch 512 dense 3.20-3.22 against
// 3.19-3.22 (there the plan changes little: 34 / 34 / 64 / 34 slabs instead of 62 / 62 / 115 / 80). 4 096 tokens per slab or
// fewer than 64 workgroups per weight lose at batch 512 (3.30-3.38 / 3.24-3.25): its four launches per layer run one after the
// other on the side stream and need the chip's width each.
static int dw_ring_plan(int64_t M, int N, int K, int* k_chunk) {
  static const int tokens = [] { const char* e = getenv("XFMR_DWR_TOKENS"); const int v = e ? atoi(e) : 3072; return v < 128 ? 128 : v; }();
  // (a function of the weight alone -- not of how many weights share the launch -- so that the grouped in-line launch and the
  //  four side-stream launches of a layer write the same slabs, bit for bit: tests/test_gpu_fullsize.py)
  static const int min_wg = [] { const char* e = getenv("XFMR_DWR_MINWG"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v; }();
  const int launch_tiles = (N / TW) * (K / TW);
  int generic_chunk;
  const int generic = xf_dw_split_plan(M, N, K, &generic_chunk);
  int64_t want = (M + tokens - 1) / tokens;
  const int64_t lo = (min_wg + launch_tiles - 1) / launch_tiles;
  if (want < lo) want = lo;
  if (want > generic) want = generic;
  if (want < 1) want = 1;
  int64_t chunk = (M + want - 1) / want;
  chunk = ((chunk + 127) / 128) * 128;
  if (chunk < generic_chunk) chunk = generic_chunk;  // (never finer than the generic plan: its slab count is the room's bound)
  *k_chunk = (int)chunk;
  return (int)((M + chunk - 1) / chunk);
}

int xf_dw_ring_launch(const XfDwItem* items, int n, int64_t M, hipStream_t st) {
  DwRingArgs p{};
  p.T = M;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const XfDwItem& t = items[i];
    int k_chunk;
    const int splits = dw_ring_plan(M, t.N, t.K, &k_chunk);
    DwRingItem& d = p.it[i];
    d.dy = reinterpret_cast<const __bf16*>(t.dy); d.x = reinterpret_cast<const __bf16*>(t.x);
    d.slabs = t.slabs; d.bias_part = t.bias_part; d.N = t.N; d.K = t.K; d.k_chunk = k_chunk; d.splits = splits;
    p.start[i] = (int)total;
    total += (int64_t)((splits + 7) / 8) * 8 * (t.N / TW) * (t.K / TW);
    if (total > 0x7fffffffll) return XFMR_EUNSUPPORTED;
    *t.splits = splits;
  }
  for (int i = n; i <= 4; ++i) p.start[i] = (int)total;
  constexpr size_t smem = (size_t)2 * NST * IMG_ELEMS * sizeof(__bf16);  // 128 KB
  {  // the dynamic-LDS limit is a per-device attribute of the function: set once per device of this process
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return XFMR_EHIP;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      if (hipFuncSetAttribute((const void*)dw_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
        return XFMR_EHIP;
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
  }
  hipLaunchKernelGGL(dw_ring_kernel, dim3((unsigned)total), dim3(512), smem, st, p);
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}

}  // extern "C"
