// Whole-encoder forward / backward: the launch sequence of BertEmbeddings + N x BertLayer over the
// kernels of this library, with a deterministic carve of the caller's activation workspace.
// Host code only (no kernels here): everything is enqueued on the caller's stream, nothing synchronises,
// so one training step can be captured into a hipGraph.
#include <math.h>
#include <stdlib.h>
#include <new>
#include <string.h>

#include "internal.h"

namespace {

struct LayerParams {
  int64_t wqkv, bqkv, wo, bo, ln1g, ln1b, w1, b1, w2, b2, ln2g, ln2b;
};
struct ParamLayout {
  int64_t pos, type, eg, eb;
  int64_t total;
};

int64_t layer_base(const xfmr_encoder_cfg* c, int i, ParamLayout* pl) {
  const int64_t H = c->hidden, I = c->inter;
  pl->pos = 0;
  pl->type = pl->pos + (int64_t)c->max_pos * H;
  pl->eg = pl->type + 2 * H;
  pl->eb = pl->eg + H;
  const int64_t first = pl->eb + H;
  const int64_t per = 3 * H * H + 3 * H + H * H + H + 2 * H + I * H + I + H * I + H + 2 * H;
  pl->total = first + per * c->layers;
  return first + per * i;
}
LayerParams layer_params(const xfmr_encoder_cfg* c, int i) {
  ParamLayout pl;
  const int64_t H = c->hidden, I = c->inter;
  int64_t o = layer_base(c, i, &pl);
  LayerParams p;
  p.wqkv = o; o += 3 * H * H;
  p.bqkv = o; o += 3 * H;
  p.wo = o; o += H * H;
  p.bo = o; o += H;
  p.ln1g = o; o += H;
  p.ln1b = o; o += H;
  p.w1 = o; o += I * H;
  p.b1 = o; o += I;
  p.w2 = o; o += H * I;
  p.b2 = o; o += H;
  p.ln2g = o; o += H;
  p.ln2b = o; o += H;
  return p;
}

// Activation storage: with the bf16 MFMA policy the tensors that are only ever MFMA operands are kept in HBM as
// bf16 ("mixed" storage; bit-identical products, half the bytes): qkv, ctx, the GELU output g, and in backward the
// gradients d_lin / d_ctx / dQKV / dI that feed the dX and dW GEMMs. f1 holds gelu'(pre) -- not the pre-activation: the forward
// epilogue has erf and exp(-x^2/2) in registers, the backward epilogue multiplies -- and is bf16 too. Everything that is added,
// normalised or reduced elementwise (residual streams, LayerNorm inputs, statistics) stays fp32.
// XFMR_ACT_FP32=1 keeps every activation fp32 (A/B measurements).
struct LayerActs {
  void *qkv, *ctx, *f1, *g;                                           // bf16 when mixed
  float *lse, *pre1, *mean1, *rstd1, *x1, *pre2, *mean2, *rstd2, *x2;  // always fp32
  void *x1b, *x2b;  // mixed storage: bf16 copies of the LayerNorm outputs (GEMM operands; x1 / x2 stay the residuals)
};
// Per-layer reduction inputs of the backward pass, reduced by ONE launch at its end (xf_multi_rowsum): split-K slabs
// of the four weight gradients, partial rows of the two bias gradients that are column sums (b1, bqkv -- produced by
// the dW GEMMs themselves), LayerNorm partial records (d gamma, d beta, and the bias gradients bo / b2).
struct RedBufs {
  float *w2, *w1, *wo, *wqkv, *b1, *bqkv, *ln2, *ln1;
};
struct Acts {
  float *emb_pre, *emb_mean, *emb_rstd, *x0;
  void* x0b;      // bf16 copy of x0 (mixed storage)
  float* emb_ln;  // embedding LayerNorm partial records
  float *dA, *dB;
  void *dLin, *dCtx, *dI, *dQKV;  // bf16 when mixed
  void* dLin2;  // in-line form: the out-proj Linear's output gradient (dLin keeps the FFN2 Linear's until the layer's
                // weight-gradient GEMMs have gone out together: xf_linear_bwd_dw_group)
  // one set PER LAYER for the side-stream dW GEMMs (xfmr_encoder_bwd): dLin in its two roles (gradient of the FFN2 / of
  // the out-proj Linear's output), dI and dQKV -- no buffer is rewritten while a weight-gradient GEMM may still read it
  void *dLinF[64], *dLinO[64], *dI2[64], *dQKV2[64];
  void* wbf;      // bf16 copy of the flat parameter buffer (mixed storage): the B operand of the forward / dX GEMMs
  void* scratch;  // ln-bwd partials / dW slabs / colsum partials (used one at a time)
  size_t scratch_bytes;
  size_t total;
};

size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// fp32 parameters -> the bf16 copy the MFMA B operands are staged from (the bf16 policy rounds them at staging time
// anyway: identical products, half the L2 -> LDS bytes of the most re-read operand)
__global__ __launch_bounds__(256) void params_to_bf16_kernel(const float* src, __bf16* dst, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) xf_st4<true>(dst, 4 * i, *reinterpret_cast<const float4*>(src + 4 * i));
}

bool mixed_storage(const xfmr_encoder_cfg* c) {
  static const bool force_fp32 = [] {
    const char* e = getenv("XFMR_ACT_FP32");
    return e && *e && *e != '0';
  }();
  return c->precision == XFMR_PREC_BF16 && !force_fp32;
}
// FFN1 -> GELU -> FFN2 -> LayerNorm as one forward kernel (gemm.hip: ffn_fwd_fused_kernel): same conditions as the
// LayerNorm-fused GEMM epilogues plus I a multiple of its chunk widths. The forward and the backward of one step must
// agree on it (f1 holds u after the fused kernel, gelu'(u) after the two-kernel form): a pure function of the
// configuration, its XFMR_ENC_*_UNFUSED flag bits included -- both calls of a step get the same cfg.
// measurement events of the call's configuration (xfmr_encoder_cfg.profile_*): event `which` of the pair, on `st`, when
// this is the part and the layer the caller named
int prof(const xfmr_encoder_cfg* c, int kind, int layer, int which, hipStream_t st) {
  if (c->profile_kernel != kind || c->profile_layer != layer || !c->profile_events[which]) return XFMR_OK;
  return hipEventRecord((hipEvent_t)c->profile_events[which], st) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}
bool ln_fused(const xfmr_encoder_cfg* c, int64_t T) {
  // out-proj / FFN2 GEMM + LayerNorm as one kernel: its 64 x 128 tiles are T / 64 workgroups -- below one per CU
  // (T < 16 384) the two-kernel form with 64 x 64 tiles is faster (batch 32: -1.5 % fused; batch 128: +0.9 %; 512: +1.8 %)
  // (round 4: from 12 288 tokens -- batch 64 x 200: 0.876 against 0.889 ms with the two LayerNorm-fused GEMMs and the FFN as
  //  separate GEMMs; at batch 32 nothing in it, 0.662 against 0.661-0.672. XFMR_LN_FUSED_MIN_TOKENS for experiments.)
  static const int64_t min_tokens = [] { const char* e = getenv("XFMR_LN_FUSED_MIN_TOKENS"); return e ? (int64_t)atoll(e) : (int64_t)12288; }();
  return mixed_storage(c) && c->hidden == 128 && T >= min_tokens && !(c->flags & XFMR_ENC_LN_UNFUSED);
}
bool ffn_fused(const xfmr_encoder_cfg* c, int64_t T) {
  // (the fused FFN pair keeps its 16 384 tokens: at 12 800 it measured 0.887 against 0.876 ms for the separate GEMMs)
  return ln_fused(c, T) && T >= 16384 && (c->inter % 128) == 0 && c->inter <= 1024 && !(c->flags & XFMR_ENC_FFN_UNFUSED);
}

// Shapes whose backward runs the weight-gradient GEMMs on the side stream (xfmr_encoder_bwd): those of the LayerNorm-fused
// dX GEMMs. The workspace holds one set of the gradient buffers per layer for them.
constexpr int64_t kDwSideTokens = 40960;
bool dw_side_shape(const xfmr_encoder_cfg* c, int64_t T) {
  // (round 2 set T >= 65 536: at batch 128 x 200 tokens the step is 1.36 ms of ~70 launches from one host thread and the 32 extra event
  //  calls cost more than the overlap gives -- 1.40 vs 1.355 ms; batch 256: even; batch 512: -2.4 %)
  if (!mixed_storage(c) || c->layers > 64) return false;
  // (round 4, with the ring weight-gradient kernel: from 40 960 tokens -- batch 256 x 200 dense 1.925-1.927 against 1.941-1.943 ms,
  //  MovieLens-like packed batches of 512 (~50 k rows) 1.912 against 1.942; packed batches of 256 (~25 k rows) 1.265 against
  //  1.237: in line below)
  return (c->flags & XFMR_ENC_DW_SIDE_ANY) || (c->hidden == 128 && T >= kDwSideTokens);
}

// Carves `base` (may be null: size query). Layer i's activations are returned in *la when i >= 0.
Acts carve(const xfmr_encoder_cfg* c, unsigned char* base, int layer, LayerActs* la, RedBufs* rb = nullptr) {
  const size_t T = (size_t)c->batch * c->seq_len, H = c->hidden, I = c->inter, A = c->heads;
  const size_t es = mixed_storage(c) ? 2 : 4;  // bytes per element of the MFMA-only tensors
  size_t o = 0;
  auto take_bytes = [&](size_t bytes) -> void* {
    void* p = base ? base + o : nullptr;
    o += up256(bytes);
    return p;
  };
  auto take = [&](size_t nfloats) -> float* { return reinterpret_cast<float*>(take_bytes(nfloats * sizeof(float))); };
  Acts a{};
  a.emb_pre = take(T * H); a.emb_mean = take(T); a.emb_rstd = take(T); a.x0 = take(T * H);
  const size_t xb = mixed_storage(c) ? T * H * 2 : 0;
  a.x0b = take_bytes(xb);
  a.dA = take(T * H); a.dB = take(T * H);
  a.dLin = take_bytes(T * H * es); a.dCtx = take_bytes(T * H * es);
  a.dLin2 = take_bytes(T * H * es);
  a.dI = take_bytes(T * I * es); a.dQKV = take_bytes(T * 3 * H * es);
  // (+236 MB per layer at T = 102 400, I = 512: only when the backward will use the side stream)
  const bool per_layer = dw_side_shape(c, (int64_t)T) && c->context && !(c->flags & XFMR_ENC_DW_INLINE);
  for (int i = 0; i < c->layers && i < 64; ++i) {
    a.dLinF[i] = (per_layer && i) ? take_bytes(T * H * es) : a.dLin;
    a.dLinO[i] = per_layer ? take_bytes(T * H * es) : a.dLin;
    a.dI2[i] = (per_layer && i) ? take_bytes(T * I * es) : a.dI;
    a.dQKV2[i] = (per_layer && i) ? take_bytes(T * 3 * H * es) : a.dQKV;
  }
  a.wbf = take_bytes(mixed_storage(c) ? (size_t)xfmr_param_count(c) * 2 : 0);
  size_t sc = xfmr_layernorm_bwd_workspace((int64_t)T, (int32_t)H);
  size_t s2 = xfmr_linear_bwd_dw_workspace((int64_t)T, (int32_t)(3 * H), (int32_t)H);
  size_t s3 = xfmr_linear_bwd_dw_workspace((int64_t)T, (int32_t)I, (int32_t)H);
  size_t s4 = xfmr_linear_bwd_dw_workspace((int64_t)T, (int32_t)H, (int32_t)I);
  size_t s5 = xfmr_linear_bwd_dw_workspace((int64_t)T, (int32_t)H, (int32_t)H);
  size_t s6 = xfmr_colsum_workspace((int64_t)T, (int32_t)(3 * H));
  size_t s7 = xfmr_colsum_workspace((int64_t)T, (int32_t)I);
  if (s2 > sc) sc = s2; if (s3 > sc) sc = s3; if (s4 > sc) sc = s4; if (s5 > sc) sc = s5;
  if (s6 > sc) sc = s6; if (s7 > sc) sc = s7;
  a.scratch = base ? base + o : nullptr;
  a.scratch_bytes = sc;
  o += up256(sc);
  {
    size_t rec = xfmr_layernorm_bwd_workspace((int64_t)T, (int32_t)H) / sizeof(float);
    const size_t rec_fused = (size_t)xf_ln_row_tiles((int64_t)T) * 3 * H;  // one record per row tile of the fused dX GEMM
    a.emb_ln = take(rec_fused > rec ? rec_fused : rec);
  }
  for (int i = 0; i < c->layers; ++i) {
    RedBufs r;
    r.w2 = take(xf_linear_bwd_dw_slab_bytes((int64_t)T, (int32_t)H, (int32_t)I) / sizeof(float));
    r.w1 = take(xf_linear_bwd_dw_slab_bytes((int64_t)T, (int32_t)I, (int32_t)H) / sizeof(float));
    r.wo = take(xf_linear_bwd_dw_slab_bytes((int64_t)T, (int32_t)H, (int32_t)H) / sizeof(float));
    r.wqkv = take(xf_linear_bwd_dw_slab_bytes((int64_t)T, (int32_t)(3 * H), (int32_t)H) / sizeof(float));
    r.b1 = take(256 * I);      // <= 256 splits (dw_split_plan)
    r.bqkv = take(256 * 3 * H);
    // LayerNorm partial records [blocks][3][H]: from the LayerNorm backward kernel, or one per 64-row tile from
    // the dX GEMM that applies the LayerNorm backward in its epilogue
    size_t lnrec = xfmr_layernorm_bwd_workspace((int64_t)T, (int32_t)H) / sizeof(float);
    const size_t lnrec_fused = (size_t)xf_ln_row_tiles((int64_t)T) * 3 * H;
    if (lnrec_fused > lnrec) lnrec = lnrec_fused;
    r.ln2 = take(lnrec);
    r.ln1 = take(lnrec);
    if (i == layer && rb) *rb = r;
  }
  for (int i = 0; i < c->layers; ++i) {
    LayerActs l;
    l.qkv = take_bytes(T * 3 * H * es); l.lse = take((size_t)c->batch * A * c->seq_len);
    l.ctx = take_bytes(T * H * es);
    l.pre1 = take(T * H); l.mean1 = take(T); l.rstd1 = take(T); l.x1 = take(T * H);
    l.f1 = take_bytes(T * I * es); l.g = take_bytes(T * I * es);
    l.pre2 = take(T * H); l.mean2 = take(T); l.rstd2 = take(T); l.x2 = take(T * H);
    l.x1b = take_bytes(xb); l.x2b = take_bytes(i + 1 < c->layers ? xb : 0);
    if (i == layer && la) *la = l;
  }
  a.total = o;
  return a;
}

int check_cfg(const xfmr_encoder_cfg* c) {
  if (!c) return XFMR_EINVAL;
  if (c->batch <= 0 || c->seq_len <= 0 || c->hidden <= 0 || c->heads <= 0 || c->inter <= 0 || c->layers <= 0)
    return XFMR_EINVAL;
  if (c->seq_len > c->max_pos) return XFMR_EINVAL;
  if ((c->hidden != c->heads * 32 && c->hidden != c->heads * 64) || (c->inter & 3)) return XFMR_EUNSUPPORTED;
  if (c->precision != XFMR_PREC_F32 && c->precision != XFMR_PREC_BF16) return XFMR_EINVAL;
  if (c->flags & ~(uint32_t)XFMR_ENC_FLAGS_ALL) return XFMR_EINVAL;  // unknown flag bits
  if (c->profile_kernel < XFMR_PROF_NONE || c->profile_kernel > XFMR_PROF_REDUCE) return XFMR_EINVAL;
  if (c->seq_offsets) {  // packed rows (ABI 3)
    if (!c->row_pos || c->packed_rows <= 0 || c->packed_rows > (int64_t)c->batch * c->seq_len) return XFMR_EINVAL;
    // the kernels that walk a sequence by its offsets: the bf16 policy's one-workgroup attention forms (head size 32, causal,
    // seq_len <= 512); everything else is row-wise and does not care
    if (!mixed_storage(c) || c->hidden != c->heads * 32 || c->seq_len > 512 || (c->flags & XFMR_ENC_BIDIRECTIONAL))
      return XFMR_EUNSUPPORTED;
  } else if (c->row_pos || c->packed_rows) {
    return XFMR_EINVAL;
  }
  return XFMR_OK;
}

enum { SITE_EMB = 0 };
inline uint32_t site_attn(int i) { return 1 + 4 * (uint32_t)i; }
inline uint32_t site_out(int i) { return 2 + 4 * (uint32_t)i; }
inline uint32_t site_ffn(int i) { return 3 + 4 * (uint32_t)i; }

#define XF_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != XFMR_OK) return _rc; \
  } while (0)

}  // namespace

extern "C" {

const char* xfmr_strerror(int code) {
  switch (code) {
    case XFMR_OK: return "ok";
    case XFMR_EINVAL: return "invalid argument";
    case XFMR_EUNSUPPORTED:
      return "shape not supported by the gfx950 kernels (head size must be 32 or 64; the fused loss takes any d_model that is a "
             "multiple of 32 up to 1024 in the bf16 policy and up to 512 in fp32; attention panels must fit LDS: head size 32 -- "
             "L <= 256 in the fp32 policy, L <= 1024 in bf16; head size 64 -- L <= 128 in fp32, L <= 256 in bf16)";
    case XFMR_EWORKSPACE: return "workspace too small";
    case XFMR_EHIP: return "HIP launch failed";
    case XFMR_EALIGN: return "pointer or leading dimension not 16-byte aligned";
    case XFMR_ECOMM: return "RCCL not loadable or an RCCL call failed (xfmr_comm_last_error)";
    default: return "unknown error";
  }
}
int xfmr_abi_version(void) { return XFMR_ABI_VERSION; }

int64_t xfmr_param_count(const xfmr_encoder_cfg* cfg) {
  if (!cfg || cfg->layers <= 0) return XFMR_EINVAL;
  ParamLayout pl;
  layer_base(cfg, 0, &pl);
  return pl.total;
}

int32_t xfmr_param_offsets(const xfmr_encoder_cfg* cfg, int64_t* offsets, int32_t capacity) {
  if (!cfg || !offsets) return XFMR_EINVAL;
  const int32_t n = 4 + 16 * cfg->layers;
  if (capacity < n) return XFMR_EINVAL;
  ParamLayout pl;
  layer_base(cfg, 0, &pl);
  const int64_t H = cfg->hidden;
  int k = 0;
  offsets[k++] = pl.pos; offsets[k++] = pl.type; offsets[k++] = pl.eg; offsets[k++] = pl.eb;
  for (int i = 0; i < cfg->layers; ++i) {
    const LayerParams p = layer_params(cfg, i);
    offsets[k++] = p.wqkv; offsets[k++] = p.wqkv + H * H; offsets[k++] = p.wqkv + 2 * H * H;
    offsets[k++] = p.bqkv; offsets[k++] = p.bqkv + H; offsets[k++] = p.bqkv + 2 * H;
    offsets[k++] = p.wo; offsets[k++] = p.bo; offsets[k++] = p.ln1g; offsets[k++] = p.ln1b;
    offsets[k++] = p.w1; offsets[k++] = p.b1; offsets[k++] = p.w2; offsets[k++] = p.b2;
    offsets[k++] = p.ln2g; offsets[k++] = p.ln2b;
  }
  return n;
}

size_t xfmr_encoder_workspace_bytes(const xfmr_encoder_cfg* cfg) {
  if (check_cfg(cfg)) return 0;
  return carve(cfg, nullptr, -1, nullptr).total;
}

// The caller-owned side stream of xfmr_encoder_bwd's weight-gradient GEMMs + the events of its fork / join.
struct XfContext {
  hipStream_t side;
  hipEvent_t ev_in, ev_done;
};
int xfmr_context_create(void** context) {
  if (!context) return XFMR_EINVAL;
  XfContext* c = new (std::nothrow) XfContext{};
  if (!c) return XFMR_EHIP;
  int lo = 0, hi = 0;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
      hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, lo) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) {
    xfmr_context_destroy(c);
    return XFMR_EHIP;
  }
  *context = c;
  return XFMR_OK;
}
int xfmr_context_destroy(void* context) {
  if (!context) return XFMR_EINVAL;
  XfContext* c = (XfContext*)context;
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return XFMR_OK;
}

int xfmr_encoder_fwd(const xfmr_encoder_cfg* cfg, const float* params, const int64_t* item_idx,
                     const float* table, int64_t n_rows, float* tok, uint8_t* key_mask, void* acts,
                     size_t acts_bytes, void* stream) {
  XF_TRY(check_cfg(cfg));
  if (!params || !item_idx || !table || !tok || !key_mask || !acts) return XFMR_EINVAL;
  if (!xf_aligned16(params) || !xf_aligned16(acts) || !xf_aligned16(tok) || !xf_aligned16(table)) return XFMR_EALIGN;
  unsigned char* base = (unsigned char*)acts;
  const Acts a = carve(cfg, base, -1, nullptr);
  if (acts_bytes < a.total) return XFMR_EWORKSPACE;
  const int B = cfg->batch, L = cfg->seq_len, H = cfg->hidden, I = cfg->inter, A = cfg->heads;
  // Tplan: what the workspace is carved for and the fusion decisions are made with (batch x seq_len, the same in the forward
  // and the backward of a step); T: the rows the row-wise kernels run -- fewer in the packed layout
  const int64_t Tplan = (int64_t)B * L;
  const int32_t* const offs = cfg->seq_offsets;
  const int64_t T = offs ? cfg->packed_rows : Tplan;
  const int prec = cfg->precision;
  const bool mix = mixed_storage(cfg);
  const bool causal = !(cfg->flags & XFMR_ENC_BIDIRECTIONAL);
  hipStream_t st = (hipStream_t)stream;
  const XfSeed sd(cfg->seed, cfg->step_device);
  ParamLayout pl;
  layer_base(cfg, 0, &pl);
  if (offs) {
    XF_TRY(xf_embed_ln_fwd_packed_ex(item_idx, table, n_rows, params + pl.pos, params + pl.type, params + pl.eg,
                                     params + pl.eb, a.x0, mix ? a.x0b : nullptr, a.emb_pre, a.emb_mean, a.emb_rstd, key_mask,
                                     T, cfg->row_pos, H, cfg->ln_eps, cfg->hidden_dropout, sd, SITE_EMB, st));
  } else {
    XF_TRY(xf_embed_ln_fwd_ex(item_idx, table, n_rows, params + pl.pos, params + pl.type, params + pl.eg,
                              params + pl.eb, a.x0, mix ? a.x0b : nullptr, a.emb_pre, a.emb_mean, a.emb_rstd, key_mask, B,
                              L, H, cfg->ln_eps, cfg->hidden_dropout, sd, SITE_EMB, st));
  }
  if (cfg->embed_event && hipEventRecord((hipEvent_t)cfg->embed_event, st) != hipSuccess) return XFMR_EHIP;  // key_mask is written
  const float* x = a.x0;
  const void* xg = mix ? a.x0b : (const void*)a.x0;  // the same activations as the GEMM operand
  const uint32_t sA = mix ? XF_S16_A : 0;
  if (mix) {
    const int64_t n4 = pl.total / 4;  // (every tensor size is a multiple of 4: H, I multiples of 32)
    hipLaunchKernelGGL(params_to_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, params,
                       (__bf16*)a.wbf, n4);
    XF_LAUNCH_CHECK();
  }
  const uint32_t sB = mix ? XF_S16_B : 0;
  const bool fuse_ln = ln_fused(cfg, Tplan);  // LayerNorm in the out-proj / FFN2 GEMM epilogues
  const bool fuse_ffn = ffn_fused(cfg, Tplan);
  auto W = [&](int64_t off) -> const float* {  // weight operand: the bf16 copy under mixed storage
    return mix ? reinterpret_cast<const float*>((const __bf16*)a.wbf + off) : params + off;
  };
  for (int i = 0; i < cfg->layers; ++i) {
    LayerActs l;
    carve(cfg, base, i, &l);
    const LayerParams p = layer_params(cfg, i);
    float* out = (i == cfg->layers - 1) ? tok : l.x2;
    XF_TRY(xf_linear_fwd_ex(xg, W(p.wqkv), params + p.bqkv, l.qkv, T, 3 * H, H, XFMR_EPI_BIAS, nullptr, nullptr,
                            0.f, 0, 0, prec, (mix ? XF_S16_C : 0) | sA | sB, st));
    XF_TRY(prof(cfg, XFMR_PROF_ATTN_FWD, i, 0, st));
    XF_TRY(xf_attn_fwd_ex(l.qkv, key_mask, l.ctx, l.lse, B, L, A, H, cfg->attn_dropout, sd, site_attn(i), prec,
                          mix, causal, st, offs));
    XF_TRY(prof(cfg, XFMR_PROF_ATTN_FWD, i, 1, st));
    if (fuse_ln) {  // LayerNorm in the GEMM epilogue (the tile spans whole rows)
      XF_TRY(xf_linear_ln_fwd_ex(l.ctx, W(p.wo), params + p.bo, l.pre1, T, H, H, x, cfg->hidden_dropout, sd,
                                 site_out(i), params + p.ln1g, params + p.ln1b, cfg->ln_eps, l.x1, l.x1b, l.mean1,
                                 l.rstd1, prec, XF_S16_A | sB, st));
    } else {
      XF_TRY(xf_linear_fwd_ex(l.ctx, W(p.wo), params + p.bo, l.pre1, T, H, H, XFMR_EPI_BIAS_DROP_RES, x, nullptr,
                              cfg->hidden_dropout, sd, site_out(i), prec, (mix ? XF_S16_A : 0) | sB, st));
      XF_TRY(xf_layernorm_fwd_ex(l.pre1, params + p.ln1g, params + p.ln1b, l.x1, mix ? l.x1b : nullptr, l.mean1,
                                 l.rstd1, T, H, cfg->ln_eps, st));
    }
    const bool last = i == cfg->layers - 1;
    XF_TRY(prof(cfg, XFMR_PROF_FFN_FWD, i, 0, st));
    if (fuse_ffn) {  // FFN1 -> GELU -> FFN2 -> dropout + residual + LayerNorm in one kernel; f1 <- the PRE-activation, g <- gelu
      XF_TRY(xf_ffn_fwd_fused_ex(l.x1b, W(p.w1), params + p.b1, W(p.w2), params + p.b2, l.f1, l.g, l.pre2, T, H, I, l.x1,
                                 cfg->hidden_dropout, sd, site_ffn(i), params + p.ln2g, params + p.ln2b,
                                 cfg->ln_eps, out, last ? nullptr : l.x2b, l.mean2, l.rstd2, st));
      XF_TRY(prof(cfg, XFMR_PROF_FFN_FWD, i, 1, st));
      x = out;
      xg = mix ? (const void*)l.x2b : (const void*)out;
      continue;
    }
    XF_TRY(xf_linear_fwd_ex(mix ? (const void*)l.x1b : (const void*)l.x1, W(p.w1), params + p.b1, l.g, T, I, H,
                            XFMR_EPI_BIAS_GELU, nullptr, l.f1, 0.f, 0, 0, prec,
                            (mix ? XF_S16_C : 0) | sA | sB | XF_AUX_GELU_GRAD, st));  // f1 <- gelu'(pre)
    if (fuse_ln) {
      XF_TRY(xf_linear_ln_fwd_ex(l.g, W(p.w2), params + p.b2, l.pre2, T, H, I, l.x1, cfg->hidden_dropout, sd,
                                 site_ffn(i), params + p.ln2g, params + p.ln2b, cfg->ln_eps, out,
                                 last ? nullptr : l.x2b, l.mean2, l.rstd2, prec, XF_S16_A | sB, st));
    } else {
      XF_TRY(xf_linear_fwd_ex(l.g, W(p.w2), params + p.b2, l.pre2, T, H, I, XFMR_EPI_BIAS_DROP_RES, l.x1, nullptr,
                              cfg->hidden_dropout, sd, site_ffn(i), prec, (mix ? XF_S16_A : 0) | sB, st));
      XF_TRY(xf_layernorm_fwd_ex(l.pre2, params + p.ln2g, params + p.ln2b, out, (mix && !last) ? l.x2b : nullptr,
                                 l.mean2, l.rstd2, T, H, cfg->ln_eps, st));
    }
    XF_TRY(prof(cfg, XFMR_PROF_FFN_FWD, i, 1, st));
    x = out;
    xg = mix ? (const void*)l.x2b : (const void*)out;
  }
  return XFMR_OK;
}

int xfmr_encoder_bwd(const xfmr_encoder_cfg* cfg, const float* params, float* grads, float* d_tok,
                     const uint8_t* key_mask, void* acts, size_t acts_bytes, void* stream) {
  XF_TRY(check_cfg(cfg));
  if (!params || !grads || !d_tok || !key_mask || !acts) return XFMR_EINVAL;
  if (!xf_aligned16(params) || !xf_aligned16(grads) || !xf_aligned16(acts) || !xf_aligned16(d_tok)) return XFMR_EALIGN;
  unsigned char* base = (unsigned char*)acts;
  const Acts a = carve(cfg, base, -1, nullptr);
  if (acts_bytes < a.total) return XFMR_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const XfSeed sd(cfg->seed, cfg->step_device);
  const int B = cfg->batch, L = cfg->seq_len, H = cfg->hidden, I = cfg->inter, A = cfg->heads;
  const int64_t Tplan = (int64_t)B * L;  // (see xfmr_encoder_fwd)
  const int32_t* const offs = cfg->seq_offsets;
  const int64_t T = offs ? cfg->packed_rows : Tplan;
  const int prec = cfg->precision;
  const bool hdrop = cfg->hidden_dropout > 0.f;
  const bool mix = mixed_storage(cfg);
  const bool causal = !(cfg->flags & XFMR_ENC_BIDIRECTIONAL);
  const uint32_t sA = mix ? XF_S16_A : 0, sC = mix ? XF_S16_C : 0, sP = mix ? XF_S16_P : 0,
                 sAB = mix ? (XF_S16_A | XF_S16_B) : 0, sB = mix ? XF_S16_B : 0;
  auto W = [&](int64_t off) -> const float* {  // weight operand: the bf16 copy the forward pass made
    return mix ? reinterpret_cast<const float*>((const __bf16*)a.wbf + off) : params + off;
  };
  const XfDropout off = xf_make_dropout(0.f, 0, 0);
  float* dX = d_tok;  // gradient w.r.t. the current layer's output
  // Two reduction lists when the caller wants the UPPER half of the flat gradient early (cfg->grads_half_event): the
  // tensors of layers >= layers / 2 are the contiguous tail of the buffer (xfmr_param_half_offset); their slabs and
  // records are reduced as soon as layer layers / 2 has enqueued its last producer, the event is recorded behind that
  // launch, and the data-parallel exchange of that half runs underneath the lower layers' backward.
  XfReduceSeg segs[12 * 64 + 2], segs_hi[12 * 64 + 2];
  int nseg = 0, nseg_hi = 0;
  if (cfg->layers > 64) return XFMR_EUNSUPPORTED;
  const int half_layer = ((cfg->grads_half_event || (cfg->flags & XFMR_ENC_REDUCE_HALF_EARLY)) && cfg->layers >= 2)
                             ? cfg->layers / 2 : -1;
  const float* const hi_begin = half_layer >= 0 ? grads + layer_params(cfg, half_layer).wqkv : nullptr;
  auto seg = [&](const float* src, float* dst, int rows, int64_t cols, int64_t ld) {
    if (hi_begin && dst >= hi_begin) segs_hi[nseg_hi++] = XfReduceSeg{src, dst, rows, (int)cols, (int)ld, 0};
    else segs[nseg++] = XfReduceSeg{src, dst, rows, (int)cols, (int)ld, 0};
  };
  // LayerNorm backward in the epilogue of the dX GEMM that produces its input gradient (whole-row 64 x 128 tiles: same
  // conditions as the forward fusion): LN1 with the FFN1 dX GEMM of its layer, LN2 of layer i-1 with the QKV dX GEMM of
  // layer i. The top layer's LN2 and the embedding LayerNorm keep their own launches.
  const bool fuse_lnb = ln_fused(cfg, Tplan);
  const bool fuse_ffn = ffn_fused(cfg, Tplan);  // what the forward of this step did
  const bool no_ffn_bwd = (cfg->flags & XFMR_ENC_FFN_BWD_UNFUSED) != 0;
  // The weight-gradient GEMMs (16 of the backward's launches, 0.49 ms at batch 512) run on the LOW-PRIORITY side stream of
  // the caller's xfmr_context: the dX -> LayerNorm -> attention chain keeps the CUs it wants, and the dW workgroups fill
  // what its 64-row-tile kernels leave idle in their last rounds (section 4 of DESIGN.md). Same priority was measured in
  // round 1 and gained nothing (each side slowed by what the overlap gave). Dependencies: a dW GEMM starts after an event
  // recorded behind the producers of its operands; the gradient buffers the chain used to reuse layer after layer (dLin in
  // both roles, dI, dQKV) exist once PER LAYER in this mode, so nothing a dW GEMM reads is rewritten before the chain
  // joins the side stream in front of the reduction launch (with two sets alternating by layer parity and
  // write-after-read events the chain kept stalling on the lagging side stream: 0.4 % instead of 2 %).
  // No context, or XFMR_ENC_DW_INLINE: everything on `st`.
  XfContext* const ctx = (XfContext*)cfg->context;
  // (the per-layer gradient buffers make this independent of the LayerNorm-fused forms: a dW GEMM only ever reads dLinF /
  //  dLinO / dI / dQKV of ITS layer and activations of the forward)
  // (packed rows: the workspace has the per-layer buffers whenever the PADDED size asks for them; whether the side stream pays
  //  is a question of the rows actually run -- ~50 000 packed rows of a MovieLens-like batch of 512: 1.915 in line against
  //  1.95 ms on the side stream, like a dense batch of 256)
  const bool dw_side = ctx && dw_side_shape(cfg, Tplan) && !(cfg->flags & XFMR_ENC_DW_INLINE) &&
                       ((cfg->flags & XFMR_ENC_DW_SIDE_ANY) || T >= kDwSideTokens || !offs);
  hipStream_t const side = dw_side ? ctx->side : nullptr;
  hipEvent_t const ev_in = dw_side ? ctx->ev_in : nullptr, ev_done = dw_side ? ctx->ev_done : nullptr;
  bool side_used = false;
  int side_rc = XFMR_OK;
  auto dw_stream = [&]() -> hipStream_t {  // everything enqueued on `st` so far is visible to the side stream
    if (!dw_side) return st;
    if (hipEventRecord(ev_in, st) != hipSuccess || hipStreamWaitEvent(side, ev_in, 0) != hipSuccess) side_rc = XFMR_EHIP;
    side_used = true;
    return side;
  };
  const bool pair_dw = !(cfg->flags & XFMR_ENC_DW_UNPAIRED);
  bool ln2_done = false;  // layer i's LN2 backward already ran inside layer i+1's QKV dX GEMM
  bool emb_ln_done = false;  // ... and the embedding LayerNorm's inside layer 0's
  // (a lambda so that a failing launch still reaches the join below: the side stream's GEMMs read the caller's buffers)
  const int chain_rc = [&]() -> int {
  for (int i = cfg->layers - 1; i >= 0; --i) {
    LayerActs l;
    RedBufs r;
    carve(cfg, base, i, &l, &r);
    const LayerParams p = layer_params(cfg, i);
    LayerActs prev;
    const void* x_in_g = mix ? a.x0b : (const void*)a.x0;  // the dW operand: the bf16 copy under mixed storage
    if (i > 0) {
      carve(cfg, base, i - 1, &prev);
      x_in_g = mix ? prev.x2b : (const void*)prev.x2;
    }
    int blocks = 0, splits = 0;
    void* const dLinF = dw_side ? a.dLinF[i] : a.dLin;   // gradient of the FFN2 Linear's output (dropout-scaled d(pre2))
    void* const dLinO = dw_side ? a.dLinO[i] : a.dLin2;  // gradient of the out-proj Linear's output
    void* const dI = dw_side ? a.dI2[i] : a.dI;
    void* const dQKV = dw_side ? a.dQKV2[i] : a.dQKV;
    // LayerNorm 2 -> dA = d(pre2); d_lin = gradient of the FFN output Linear (dropout-scaled copy of it)
    const bool lin_copy = hdrop || mix;  // without dropout and with fp32 storage d_lin IS dx
    if (!ln2_done) {
      XF_TRY(xf_layernorm_bwd_impl(dX, l.pre2, l.mean2, l.rstd2, params + p.ln2g, a.dA, lin_copy ? dLinF : nullptr,
                                   mix, nullptr, nullptr, nullptr, T, H, off,
                                   xf_make_dropout(cfg->hidden_dropout, sd, site_ffn(i)), r.ln2, st, &blocks));
      seg(r.ln2, grads + p.ln2g, blocks, H, 3 * H);
      seg(r.ln2 + H, grads + p.ln2b, blocks, H, 3 * H);
      seg(r.ln2 + 2 * H, grads + p.b2, blocks, H, 3 * H);
    }
    const void* dlin = lin_copy ? dLinF : (const void*)a.dA;
    const bool fuse_ffn_bwd = fuse_ffn && fuse_lnb && !no_ffn_bwd;
    // In line (no side stream) the four weight-gradient GEMMs of the layer go out in ONE launch once its last operand
    // (dQKV) exists (xf_linear_bwd_dw_group; same slabs bit for bit): 16 launches of ~8 us become 4 at batch 32. Their
    // operands all live to the end of the layer: dLinF (dLin), dI, dLinO (dLin2), dQKV are four different buffers and the
    // next writer of any of them is the next layer. On the side stream each GEMM keeps its own launch: there the FFN2 one
    // starts underneath the FFN dX kernel, long before dQKV exists.
    // (grouped on the side stream too: 3.245-3.265 against 3.244-3.269 ms/step at batch 512, 6.09-6.16 against 6.07-6.10 at 1024)
    const bool group_dw = pair_dw && !dw_side && mix;  // (bf16 storage: the dLin copies exist; fp32 keeps one launch each)
    int splits_w2 = 0, splits_w1 = 0, splits_wo = 0;
    if (!group_dw) {
      XF_TRY(xf_linear_bwd_dw_deferred(dlin, l.g, T, H, I, prec, sAB, r.w2, nullptr, &splits, dw_stream()));
      seg(r.w2, grads + p.w2, splits, (int64_t)H * I, (int64_t)H * I);
    }
    const void* const dlin_ffn = dlin;
    XF_TRY(prof(cfg, XFMR_PROF_FFN_BWD, i, 0, st));  // (in the unfused forms: FFN2 dX ... LayerNorm 1 backward on `st`)
    if (fuse_ffn_bwd) {  // FFN2 dX * gelu'(u) -> dI -> FFN1 dX (+= d(pre2)) -> LayerNorm 1 backward in one kernel
      XF_TRY(xf_ffn_bwd_dx_fused_ex(dlin, W(p.w2), l.f1, W(p.w1), dI, T, H, I, a.dA, l.pre1, l.mean1, l.rstd1,
                                    params + p.ln1g, cfg->hidden_dropout, sd, site_out(i), dX, dLinO, r.ln1,
                                    &blocks, st));
    } else {
      // (after the fused FFN forward f1 holds the pre-activation u, not gelu'(u): the epilogue evaluates gelu'(u))
      XF_TRY(xf_linear_bwd_dx_ex(dlin, W(p.w2), dI, T, H, I, nullptr, l.f1, prec,
                                 sA | sC | sP | sB | (fuse_ffn ? 0 : XF_AUX_GELU_GRAD), st));
    }
    if (!group_dw) {
      XF_TRY(xf_linear_bwd_dw_deferred(dI, mix ? (const void*)l.x1b : (const void*)l.x1, T, I, H, prec, sAB, r.w1, r.b1, &splits, dw_stream()));  // + b1 partial rows
      seg(r.w1, grads + p.w1, splits, (int64_t)I * H, (int64_t)I * H);
      seg(r.b1, grads + p.b1, splits, I, I);
    }
    if (fuse_ffn_bwd) {  // (done above)
    } else if (fuse_lnb) {  // dX of FFN1 (+= d(pre2)) and LayerNorm 1 backward in one kernel -> dX = d(pre1), dLin
      XF_TRY(xf_linear_bwd_dx_lnbwd_ex(dI, W(p.w1), T, I, H, a.dA, l.pre1, l.mean1, l.rstd1, params + p.ln1g,
                                       cfg->hidden_dropout, sd, site_out(i), dX, dLinO, r.ln1, &blocks, prec,
                                       sA | sB, st));
    } else {
      XF_TRY(xf_linear_bwd_dx_ex(dI, W(p.w1), a.dA, T, I, H, a.dA, nullptr, prec, sA | sB, st));  // += d(pre2)
      // LayerNorm 1 -> dX = d(pre1)
      XF_TRY(xf_layernorm_bwd_impl(a.dA, l.pre1, l.mean1, l.rstd1, params + p.ln1g, dX, lin_copy ? dLinO : nullptr,
                                   mix, nullptr, nullptr, nullptr, T, H, off,
                                   xf_make_dropout(cfg->hidden_dropout, sd, site_out(i)), r.ln1, st, &blocks));
    }
    XF_TRY(prof(cfg, XFMR_PROF_FFN_BWD, i, 1, st));
    seg(r.ln1, grads + p.ln1g, blocks, H, 3 * H);
    seg(r.ln1 + H, grads + p.ln1b, blocks, H, 3 * H);
    seg(r.ln1 + 2 * H, grads + p.bo, blocks, H, 3 * H);
    dlin = lin_copy ? dLinO : (const void*)dX;
    if (!group_dw) {
      XF_TRY(xf_linear_bwd_dw_deferred(dlin, l.ctx, T, H, H, prec, sAB, r.wo, nullptr, &splits, dw_stream()));
      seg(r.wo, grads + p.wo, splits, (int64_t)H * H, (int64_t)H * H);
    }
    XF_TRY(xf_linear_bwd_dx_ex(dlin, W(p.wo), a.dCtx, T, H, H, nullptr, nullptr, prec, sA | sC | sB, st));  // d(ctx)
    XF_TRY(prof(cfg, XFMR_PROF_ATTN_BWD, i, 0, st));
    XF_TRY(xf_attn_bwd_ex(l.qkv, key_mask, l.ctx, l.lse, a.dCtx, dQKV, B, L, A, H, cfg->attn_dropout, sd,
                          site_attn(i), prec, mix, causal, st, offs));
    XF_TRY(prof(cfg, XFMR_PROF_ATTN_BWD, i, 1, st));
    if (group_dw) {
      const XfDwItem items[4] = {
          {dlin_ffn, l.g, (int32_t)H, (int32_t)I, r.w2, nullptr, &splits_w2},
          {dI, mix ? (const void*)l.x1b : (const void*)l.x1, (int32_t)I, (int32_t)H, r.w1, r.b1, &splits_w1},
          {dlin, l.ctx, (int32_t)H, (int32_t)H, r.wo, nullptr, &splits_wo},
          {dQKV, x_in_g, (int32_t)(3 * H), (int32_t)H, r.wqkv, r.bqkv, &splits}};
      XF_TRY(prof(cfg, XFMR_PROF_DW, i, 0, st));  // (the in-line form: the layer's four weight-gradient GEMMs are ONE launch)
      XF_TRY(xf_linear_bwd_dw_group(items, 4, T, prec, sAB, st));
      XF_TRY(prof(cfg, XFMR_PROF_DW, i, 1, st));
      seg(r.w2, grads + p.w2, splits_w2, (int64_t)H * I, (int64_t)H * I);
      seg(r.w1, grads + p.w1, splits_w1, (int64_t)I * H, (int64_t)I * H);
      seg(r.b1, grads + p.b1, splits_w1, I, I);
      seg(r.wo, grads + p.wo, splits_wo, (int64_t)H * H, (int64_t)H * H);
    } else {
      XF_TRY(xf_linear_bwd_dw_deferred(dQKV, x_in_g, T, 3 * H, H, prec, sAB, r.wqkv, r.bqkv, &splits, dw_stream()));
    }
    seg(r.wqkv, grads + p.wqkv, splits, (int64_t)3 * H * H, (int64_t)3 * H * H);
    seg(r.bqkv, grads + p.bqkv, splits, 3 * H, 3 * H);
    ln2_done = false;
    if (fuse_lnb && i > 0) {  // dX of QKV (+= d(pre1)) and layer i-1's LayerNorm 2 backward -> dA = d(pre2), dLin
      RedBufs rp;
      carve(cfg, base, i - 1, &prev, &rp);
      const LayerParams pp = layer_params(cfg, i - 1);
      XF_TRY(xf_linear_bwd_dx_lnbwd_ex(dQKV, W(p.wqkv), T, 3 * H, H, dX, prev.pre2, prev.mean2, prev.rstd2,
                                       params + pp.ln2g, cfg->hidden_dropout, sd, site_ffn(i - 1), a.dA,
                                       dw_side ? a.dLinF[i - 1] : a.dLin, rp.ln2, &blocks, prec, sA | sB, st));
      seg(rp.ln2, grads + pp.ln2g, blocks, H, 3 * H);
      seg(rp.ln2 + H, grads + pp.ln2b, blocks, H, 3 * H);
      seg(rp.ln2 + 2 * H, grads + pp.b2, blocks, H, 3 * H);
      ln2_done = true;
    } else if (fuse_lnb) {  // layer 0: dX of QKV (+= d(pre1)) and the EMBEDDING LayerNorm backward -> dA
      ParamLayout pe;
      layer_base(cfg, 0, &pe);
      XF_TRY(xf_linear_bwd_dx_lnbwd_ex(dQKV, W(p.wqkv), T, 3 * H, H, dX, a.emb_pre, a.emb_mean, a.emb_rstd,
                                       params + pe.eg, 0.f, sd, 0, a.dA, nullptr, a.emb_ln, &blocks, prec,
                                       sA | sB, st, cfg->hidden_dropout, SITE_EMB));
      seg(a.emb_ln, grads + pe.eg, blocks, H, 3 * H);
      seg(a.emb_ln + H, grads + pe.eb, blocks, H, 3 * H);
      emb_ln_done = true;
    } else {
      XF_TRY(xf_linear_bwd_dx_ex(dQKV, W(p.wqkv), dX, T, 3 * H, H, dX, nullptr, prec, sA | sB, st));  // += d(pre1)
    }
    if (i == half_layer) {  // every producer of the upper half's slabs / records is enqueued: finish that half now
      hipStream_t rs = dw_stream();  // (the side stream when the dW GEMMs run there: the chain itself does not wait)
      XF_TRY(xf_multi_rowsum(segs_hi, nseg_hi, rs));
      if (cfg->grads_half_event && hipEventRecord((hipEvent_t)cfg->grads_half_event, rs) != hipSuccess) return XFMR_EHIP;
    }
  }
  ParamLayout pl;
  layer_base(cfg, 0, &pl);
  if (!emb_ln_done) {
    int blocks = 0;
    XF_TRY(xf_layernorm_bwd_impl(dX, a.emb_pre, a.emb_mean, a.emb_rstd, params + pl.eg, a.dA, nullptr, false, nullptr,
                                 nullptr, nullptr, T, H, xf_make_dropout(cfg->hidden_dropout, sd, SITE_EMB), off,
                                 a.emb_ln, st, &blocks));
    seg(a.emb_ln, grads + pl.eg, blocks, H, 3 * H);
    seg(a.emb_ln + H, grads + pl.eb, blocks, H, 3 * H);
  }
  // (Measured and not kept: the weight-gradient GEMMs on a side stream beside the dX -> LayerNorm -> attention chain.
  // The kernels do overlap, and each slows down by what the overlap would have gained: 1.910 vs 1.904 ms/step.)
  // (Measured and not kept: reducing each layer's slabs and records on the side stream as soon as they are enqueued, so that
  // only layer 0's are left for the end: 3.41 vs 3.34 ms/step -- the low-priority reductions slow the chain's tail.)
  // the position / type embedding gradients need only the chain's last output: in front of the join, underneath whatever
  // the side stream still has to do
  if (offs) XF_TRY(xf_embed_param_grads_packed(a.dA, grads + pl.pos, grads + pl.type, offs, B, L, H, cfg->max_pos, st));
  else XF_TRY(xfmr_embed_param_grads(a.dA, grads + pl.pos, grads + pl.type, B, L, H, cfg->max_pos, stream));
  return XFMR_OK;
  }();
  if (side_used) {  // the chain joins the side stream: the reduction launch reads every slab
    if (hipEventRecord(ev_done, side) != hipSuccess || hipStreamWaitEvent(st, ev_done, 0) != hipSuccess) side_rc = XFMR_EHIP;
  }
  if (chain_rc != XFMR_OK) return chain_rc;
  if (side_rc != XFMR_OK) return side_rc;
  XF_TRY(prof(cfg, XFMR_PROF_REDUCE, cfg->profile_layer, 0, st));
  XF_TRY(xf_multi_rowsum(segs, nseg, st));  // every weight / bias / LayerNorm gradient of the encoder, one launch
  XF_TRY(prof(cfg, XFMR_PROF_REDUCE, cfg->profile_layer, 1, st));
  if (cfg->grads_half_event && half_layer < 0 &&  // (a one-layer encoder has no upper half: the event marks the whole buffer)
      hipEventRecord((hipEvent_t)cfg->grads_half_event, st) != hipSuccess)
    return XFMR_EHIP;
  return XFMR_OK;
}

int64_t xfmr_param_half_offset(const xfmr_encoder_cfg* cfg) {
  if (!cfg || cfg->layers <= 0) return XFMR_EINVAL;
  if (cfg->layers < 2) return 0;
  return layer_params(cfg, cfg->layers / 2).wqkv;
}

}  // extern "C"
