/*
 * xfmr_hip.h -- C ABI of libxfmr_hip.so: the MI355X (gfx950 / CDNA4) training hot path of the
 * sequential transformer recommender.
 *
 * Boundary: the reference has no FFI layer; its "operator API" for this path is
 * PyTorch nn.Module + autograd. Each entry point below therefore replaces one ATen-level compute site
 * that the reference reaches through torch / transformers, and cites it. Paths are relative to the
 * reference checkout (xfmr_rec/...) or, prefixed TF:, to the `transformers` package the reference
 * instantiates its encoder from (xfmr_rec/models.py:93-102).
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer (inputs, outputs, workspace) is owned by the caller
 *     and lives in device memory of the current HIP device; no entry point allocates, frees or retains device memory.
 *     The only objects the library creates are the ones the caller asks for and owns: streams, events
 *     (xfmr_stream_create / xfmr_event_create ...) and the xfmr_context object: a side stream + two events.
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the default stream); no compute entry
 *     point synchronises the host or creates a HIP object, so a sequence of them can be captured into a hipGraph
 *     (stream capture follows the event fork / join onto the context's side stream). What changes from step to step
 *     in a captured step -- the dropout stream and AdamW's step count -- is read from device memory:
 *     xfmr_encoder_cfg.step_device, xfmr_adamw_dev, xfmr_step_advance. (xfmr_batch_upload and xfmr_event_synchronize
 *     do wait on the host, by design; they are not part of the step's kernels.)
 *   - all floating-point tensors in HBM are fp32, row-major, contiguous unless a stride is given.
 *     `precision` selects the matrix-core arithmetic of every contraction:
 *         XFMR_PREC_F32  : v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; parity mode)
 *         XFMR_PREC_BF16 : v_mfma_f32_32x32x16_bf16 (operands rounded to bf16 on the way into LDS,
 *                          fp32 accumulate; the analogue of the reference's `bf16-mixed`,
 *                          xfmr_rec/trainer.py:450)
 *     LayerNorm, softmax, loss reductions, AdamW always run in fp32.
 *   - item indices are int64 as the reference's batches carry them (xfmr_rec/data.py:534-540).
 *   - return value: 0 on success, a negative XFMR_E* code otherwise (never throws, never aborts).
 *     Asynchronous HIP errors surface at the caller's next synchronisation.
 *   - re-entrant: no device-side state survives a call and the library keeps no per-thread or global mutable state;
 *     safe for one process per GPU and for several host threads on different streams (each with its own xfmr_context).
 *     Everything optional about a call is an argument: an event to record, "d_tok is already zero", profiling events,
 *     fusion switches are fields of the cfg structs (ABI 1 had one-shot per-thread hooks for them). A few XFMR_*
 *     environment variables (tile / split tuning only, listed in DESIGN.md section 5; none changes what a forward pass
 *     leaves for its backward) exist for experiments; every setting computes the same function to rounding, none is
 *     needed in production.
 *   - the data-parallel gradient exchange (SURVEY.md section 8b lists an `allreduce_flat` op) is deliberately NOT an
 *     entry point here: the flat gradient is one contiguous device buffer, and torch.distributed's all_reduce over
 *     RCCL (xfmr_rec_amd/distributed.py) is the exchange -- there is no kernel of ours in it to export.
 */
#ifndef XFMR_HIP_H
#define XFMR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XFMR_ABI_VERSION 3

enum {
  XFMR_OK = 0,
  XFMR_EINVAL = -1,      /* bad argument (null pointer, non-positive size)            */
  XFMR_EUNSUPPORTED = -2, /* shape outside what the kernels are built for              */
  XFMR_EWORKSPACE = -3,  /* workspace too small                                        */
  XFMR_EHIP = -4,        /* a HIP launch failed (hipGetLastError)                      */
  XFMR_EALIGN = -5,      /* pointer / leading dimension not 16-byte aligned            */
  XFMR_ECOMM = -6        /* RCCL not loadable, or an RCCL call failed (xfmr_comm_*)    */
};

enum { XFMR_PREC_F32 = 0, XFMR_PREC_BF16 = 1 };
/* Self-attention mask. CAUSAL is what the reference builds (BertConfig(is_decoder=True), xfmr_rec/models.py:355);
 * BIDIRECTIONAL is BertConfig.is_decoder=False: the key-padding mask alone (TF:masking_utils.py bidirectional mask). */
enum { XFMR_ATTN_CAUSAL = 0, XFMR_ATTN_BIDIRECTIONAL = 1 };
/* xfmr_encoder_cfg.flags. The *_UNFUSED / DW_INLINE bits select the separate-launch forms of fused kernels (A/B
 * measurements, parity of fused vs unfused forms in tests/): the forward and the backward of a step get the same cfg, so
 * they agree on what the saved activations hold. */
enum {
  XFMR_ENC_BIDIRECTIONAL = 1u,
  XFMR_ENC_LN_UNFUSED = 2u,      /* LayerNorm (and its backward) as launches of their own, not GEMM epilogues      */
  XFMR_ENC_FFN_UNFUSED = 4u,     /* FFN forward as two GEMM launches                                               */
  XFMR_ENC_FFN_BWD_UNFUSED = 8u, /* FFN backward dX chain as two GEMM launches                                     */
  XFMR_ENC_DW_INLINE = 16u,      /* weight-gradient GEMMs on the caller's stream even when cfg.context is given     */
  XFMR_ENC_DW_SIDE_ANY = 32u,    /* ... on the context's side stream at ANY size. Without it the side stream is used from
                                    65 536 tokens at d_model 128 only: below that the fork / join event calls of an eager
                                    step cost the host more than the overlap gives the GPU. In a captured hipGraph
                                    (xfmr_rec_amd.trainer.GraphedStep) those calls cost nothing at replay, and a small
                                    step's weight-gradient GEMMs run beside its latency-bound dX chain.              */
  XFMR_ENC_DW_UNPAIRED = 64u,    /* every weight-gradient GEMM a launch of its own. Without it the in-line form (no side
                                    stream, bf16 storage) launches the four of a layer together: identical slabs.       */
  XFMR_ENC_REDUCE_HALF_EARLY = 128u, /* reduce the upper layers' split-K slabs / records as soon as layer L/2 is enqueued --
                                    on the side stream when the dW GEMMs run there -- also WITHOUT grads_half_event (which
                                    implies it): the final in-chain reduction launch then reads half the slabs            */
  XFMR_ENC_FLAGS_ALL = 255u
};

/* Index tensors of the PACKED layout (xfmr_encoder_cfg.seq_offsets) from a collated padded batch: hist / pos / neg are the
 * SeqBatch's three (batch, seq_len) int64 tensors (data.py:534-540), offsets64 (batch + 1, device, int64 -- it travels in
 * the same host block as the batch) the row offsets computed on the host from the rows' lengths (len_b = seq_len minus
 * the row's trailing zeros of hist). Outputs: hist_p / pos_p / neg_p (packed_rows) = rows [0, len_b) of every sequence
 * back to back, offsets32 (batch + 1) and row_pos (packed_rows) for xfmr_encoder_cfg. One launch. */
int xfmr_pack_rows(const int64_t* hist, const int64_t* pos, const int64_t* neg, const int64_t* offsets64, int32_t batch,
                   int32_t seq_len, int64_t packed_rows, int64_t* hist_p, int64_t* pos_p, int64_t* neg_p,
                   int32_t* offsets32, int32_t* row_pos, void* stream);
/* The same with the sequences taken in another ORDER: packed slot b holds batch row order[b] (device, int64, a permutation of
 * 0 .. batch - 1; offsets64 = the cumulative lengths in that order). Nothing downstream depends on the order of the rows of a
 * batch -- every loss is a sum over them (trainer.py:250-264) --, and the one-workgroup-per-(sequence, head) attention kernels
 * take their workgroups in slot order: with the LONGEST sequences first the launch does not end on a 200-token sequence that
 * started last (MovieLens-like batches of 512, one-stream trace: attention backward 72.6 -> 58.2 us per layer, forward 26.2 ->
 * 20.2; the step 1.862-1.872 -> 1.846-1.864 ms). order == NULL: xfmr_pack_rows. */
int xfmr_pack_rows_ordered(const int64_t* hist, const int64_t* pos, const int64_t* neg, const int64_t* offsets64,
                           const int64_t* order, int32_t batch, int32_t seq_len, int64_t packed_rows, int64_t* hist_p,
                           int64_t* pos_p, int64_t* neg_p, int32_t* offsets32, int32_t* row_pos, void* stream);

/* Loss heads, in the order of the reference's LOSS_CLASSES (xfmr_rec/losses.py:546-554). */
enum {
  XFMR_LOSS_ALIGNMENT = 0,
  XFMR_LOSS_ALIGNMENT_CONTRASTIVE = 1,
  XFMR_LOSS_CONTRASTIVE = 2,
  XFMR_LOSS_INFONCE = 3,
  XFMR_LOSS_NCE = 4,
  XFMR_LOSS_PAIRWISE_HINGE = 5,
  XFMR_LOSS_PAIRWISE_LOGISTIC = 6,
  XFMR_NUM_LOSSES = 7
};

const char* xfmr_strerror(int code);
int xfmr_abi_version(void);
/* A non-blocking HIP stream of the device's LOWEST priority (hipStreamCreateWithPriority), for work that should fill
 * what the training chain leaves idle rather than compete with it: the deferred logging pass of the loss
 * (xfmr_rec_amd/trainer.py), the weight-gradient GEMMs inside xfmr_encoder_bwd. The caller owns it
 * (xfmr_stream_destroy). torch only offers "normal" and "high". */
int xfmr_low_priority_stream_create(void** stream);
int xfmr_stream_create(void** stream); /* non-blocking, default priority (the copy stream of xfmr_batch_upload) */
int xfmr_stream_destroy(void* stream);
/* hipEvent_t objects (timing != 0: usable with xfmr_event_elapsed_ms). Created once by the caller and re-recorded every
 * step: the training step creates no HIP object on its way. */
int xfmr_event_create(void** event, int32_t timing);
int xfmr_event_destroy(void* event);
int xfmr_event_record(void* event, void* stream);
int xfmr_stream_wait_event(void* stream, void* event);
int xfmr_event_elapsed_ms(void* start_event, void* stop_event, float* ms);
int xfmr_event_synchronize(void* event); /* blocks the HOST until the event has completed */
int xfmr_event_query(void* event);       /* 1 = completed, 0 = not yet, < 0 = error */

/* ------------------------------------------------------------------------------------------------
 * Host -> HBM hand-over of one collated batch: what Lightning's batch transfer does with the output of the
 * reference's pin_memory DataLoader (xfmr_rec/data.py:915-927: DataLoader(..., pin_memory=True); the SeqBatch's three
 * (B,L) int64 index tensors, data.py:534-540, 799-805). ONE hipMemcpyAsync of `bytes` from page-locked host memory into
 * a device slot on `copy_stream`, `ready_event` recorded behind it; the compute stream waits for that event with
 * xfmr_stream_wait_event in front of the step that reads the slot. `slot_free_event` (recorded by the caller on the
 * compute stream after the last kernel that read the slot; NULL = never used) guards the slot's reuse: the call waits
 * for it ON THE HOST -- which returns at once unless the host is a whole ring of slots ahead of the GPU (bounded
 * run-ahead). It is deliberately not a stream wait: a hipStreamWaitEvent in front of a copy makes hipMemcpyAsync itself
 * block the host until the event has completed (measured: 0.9 ms per step at the bench shape).
 * Nothing is allocated (xfmr_rec_amd/data.py: PinnedBatchRing keeps the slots, the copy stream and the events).
 * ---------------------------------------------------------------------------------------------- */
int xfmr_batch_upload(void* dst_device, const void* src_pinned, size_t bytes, void* copy_stream, void* slot_free_event,
                      void* ready_event);

/* ------------------------------------------------------------------------------------------------
 * Encoder configuration and the flat parameter layout.
 *
 * All trainable tensors of the causal BERT encoder live in ONE flat fp32 buffer (and their gradients
 * and AdamW moments in buffers of the same layout), so the data-parallel gradient exchange is a single
 * all-reduce and the optimizer a single launch. Tensor order (HF state_dict names, the reference's
 * checkpoint keys under `model.model.0.auto_model.`):
 *   embeddings.position_embeddings.weight (max_pos,H) | embeddings.token_type_embeddings.weight (2,H)
 *   | embeddings.LayerNorm.{weight,bias} (H) | per layer: attention.self.{query,key,value}.weight
 *   stored as one (3H,H) block followed by the (3H) bias block | attention.output.dense.{weight,bias}
 *   | attention.output.LayerNorm.{weight,bias} | intermediate.dense.{weight (I,H),bias (I)}
 *   | output.dense.{weight (H,I),bias (H)} | output.LayerNorm.{weight,bias}.
 * `word_embeddings` and `pooler.*` never receive a gradient on this path and are not
 * part of the buffer.
 * ---------------------------------------------------------------------------------------------- */
typedef struct xfmr_encoder_cfg {
  int32_t batch;      /* B: sequences in this call                                  */
  int32_t seq_len;    /* L: padded length of every sequence (<= max_pos)            */
  int32_t hidden;     /* H: d_model == item-embedding width (xfmr_rec/models.py:336-345) */
  int32_t heads;      /* A: attention heads; H/A must be 32 or 64                    */
  int32_t inter;      /* I: FFN width                                                */
  int32_t layers;     /* number of BertLayers                                        */
  int32_t max_pos;    /* rows of the position-embedding table (ModelConfig.max_seq_length) */
  int32_t precision;  /* XFMR_PREC_*                                                 */
  float ln_eps;       /* 1e-12 (TF:models/bert/configuration_bert.py:44-63)          */
  float hidden_dropout;  /* 0.1 in training (same file), 0 for eval / parity         */
  float attn_dropout;    /* 0.1 in training, 0 for eval / parity                     */
  uint32_t flags;     /* XFMR_ENC_* bits; 0 = the reference's setting (causal decoder-style mask) */
  uint64_t seed;      /* dropout stream of this step; fwd and bwd must pass the same */
  /* ---- ABI 2: everything optional about a call is an argument (all three may be NULL) ---- */
  const uint32_t* step_device; /* device counter mixed into the dropout stream ON THE DEVICE (kernel entry): a captured
                                  hipGraph replays the same `seed`, the counter (xfmr_step_advance) makes every replay
                                  a new mask. Forward and backward of a step must see the same value.               */
  void* embed_event;  /* hipEvent_t: xfmr_encoder_fwd records it on `stream` right after the launch that writes
                         key_mask -- work that needs only the mask (xfmr_sampled_loss_prepare) can run on another
                         stream underneath the rest of the forward. Ignored by xfmr_encoder_bwd.                    */
  void* context;      /* an xfmr_context handle (xfmr_context_create): xfmr_encoder_bwd runs its weight-gradient GEMMs on the
                         context's lowest-priority side stream and joins it into `stream` before its last launch;
                         NULL = everything on `stream`.                                                            */
  void* grads_half_event; /* hipEvent_t: xfmr_encoder_bwd finishes the gradients of layers >= layers / 2 -- the contiguous
                         tail of the flat buffer from xfmr_param_half_offset(cfg) on -- as soon as that layer's backward is
                         enqueued and records the event behind them: the data-parallel all-reduce of that half can run
                         underneath the lower layers' backward (xfmr_rec_amd/distributed.py). The rest of the buffer is
                         complete when the call's last launch has run, as always. Ignored by xfmr_encoder_fwd.        */
  /* Measurement (bench.py, like xfmr_loss_cfg.profile_*): two hipEvent_t recorded on `stream` in front of and behind ONE
     part of the encoder -- the kernel(s) XFMR_PROF_* names, of layer `profile_layer` -- by whichever of xfmr_encoder_fwd /
     xfmr_encoder_bwd launches it. profile_kernel 0 or a null event: nothing is recorded.                               */
  int32_t profile_kernel;
  int32_t profile_layer;
  void* profile_events[2];
  /* ---- ABI 3: PACKED rows (all zero / NULL = the padded (B, L) layout above) ----
   * The reference's collate right-pads every sequence to the batch's longest (xfmr_rec/data.py:799-805) and the encoder
   * then computes the padding rows too; nothing ever reads them (models.py:392 drops them, attention masks them as keys,
   * their output gradient is zero). With seq_offsets the token axis holds only the rows [0, len_b) of every sequence,
   * back to back: sequence b = rows [seq_offsets[b], seq_offsets[b + 1]) of item_idx / tok / key_mask / d_tok (all
   * `packed_rows` = seq_offsets[batch] rows long), len_b <= seq_len. Every row-wise kernel (Linears, LayerNorms, FFN,
   * their backward, the weight gradients) then runs packed_rows rows instead of batch * seq_len; attention walks each
   * sequence's own length. Valid rows get the values of the padded layout bit for bit (dropout aside: its masks are keyed
   * by the row index). bf16 policy, head size 32, causal, seq_len <= 256 only (XFMR_EUNSUPPORTED otherwise); workspaces
   * are sized for batch * seq_len as before. xfmr_pack_rows builds the index tensors of this layout.               */
  const int32_t* seq_offsets; /* device, batch + 1 entries, ascending, seq_offsets[0] == 0                          */
  const int32_t* row_pos;     /* device, packed_rows entries: position of each packed row within its sequence
                                 (the row of the position-embedding table it adds)                                  */
  int64_t packed_rows;        /* seq_offsets[batch], known to the host (it sizes the launches)                      */
} xfmr_encoder_cfg;
/* xfmr_encoder_cfg.profile_kernel: the FFN forward (one kernel in the fused form: FFN1 + GELU + FFN2 + dropout + residual +
 * LayerNorm), the FFN backward's dX chain (one kernel in the fused form), the attention forward, the attention backward. */
enum { XFMR_PROF_NONE = 0, XFMR_PROF_FFN_FWD = 1, XFMR_PROF_FFN_BWD = 2, XFMR_PROF_ATTN_FWD = 3, XFMR_PROF_ATTN_BWD = 4,
       XFMR_PROF_DW = 5,      /* the layer's four weight-gradient GEMMs where they are ONE launch (the in-line form:
                                 XFMR_ENC_DW_INLINE / no context; on the side stream they are four launches at four times
                                 and nothing is recorded) */
       XFMR_PROF_REDUCE = 6   /* the backward's final reduction launch (split-K slabs, bias rows, LayerNorm records) */ };
/* Element offset at which the early-finished upper half of the flat gradient begins (0 for a one-layer encoder: the
 * event then marks the whole buffer, recorded behind the last launch). */
int64_t xfmr_param_half_offset(const xfmr_encoder_cfg* cfg);

/* A caller-owned side stream of the device's LOWEST priority plus the two events of the fork / join (created on the
 * current device). One per host thread / stream that drives xfmr_encoder_bwd concurrently. */
int xfmr_context_create(void** context);
int xfmr_context_destroy(void* context);

/* Number of fp32 elements of the flat parameter buffer. */
int64_t xfmr_param_count(const xfmr_encoder_cfg* cfg);
/* Offsets (in elements) of the tensors in the order listed above: 4 + 16*layers entries
 * (q,k,v weights are 3 consecutive (H,H) entries, then 3 (H) biases). Returns the entry count. */
int32_t xfmr_param_offsets(const xfmr_encoder_cfg* cfg, int64_t* offsets, int32_t capacity);

/* ------------------------------------------------------------------------------------------------
 * K1-K3: item-embedding gather + attention-mask derivation + BertEmbeddings.
 * Replaces xfmr_rec/models.py:336-338 (torch.nn.Embedding gather), :343 (mask = (emb != 0).any(-1))
 * and TF:models/bert/modeling_bert.py:68-108 (x + type_emb[0] + pos_emb[t] -> LayerNorm -> dropout).
 *   item_idx (B*L) int64, table (n_rows,H) frozen, pos_emb (>=L,H), type_emb (2,H), gamma/beta (H)
 *   out (B*L,H) post-LN(+dropout), pre (B*L,H) LN input (saved for backward), mean/rstd (B*L),
 *   key_mask (B*L) uint8 = 1 where the gathered row has a non-zero element.
 * An index outside [0,n_rows) is a caller error; it is clamped to row 0 (padding) rather than faulting.
 * ---------------------------------------------------------------------------------------------- */
int xfmr_embed_ln_fwd(const int64_t* item_idx, const float* table, int64_t n_rows, const float* pos_emb,
                      const float* type_emb, const float* gamma, const float* beta, float* out, float* pre,
                      float* mean, float* rstd, uint8_t* key_mask, int32_t B, int32_t L, int32_t H, float eps,
                      float dropout_p, uint64_t seed, uint32_t site, void* stream);

/* Gradients of the embedding stage w.r.t. position / token-type embeddings (the item table is frozen,
 * xfmr_rec/models.py:251-253): d_pos[t] = sum_b d_pre[b,t], d_type[0] = sum_t d_pos[t], d_type[1] = 0,
 * rows >= L of d_pos are zeroed. d_pre is the output of xfmr_layernorm_bwd on the embedding LayerNorm. */
int xfmr_embed_param_grads(const float* d_pre, float* d_pos, float* d_type, int32_t B, int32_t L, int32_t H,
                           int32_t max_pos, void* stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (TF:models/bert/modeling_bert.py:292, :349 -- torch.nn.LayerNorm over the last dim).
 * fwd: y = (x-mean)*rstd*gamma+beta; saves mean, rstd.
 * bwd: dx (rows,H); per-column sums d_gamma, d_beta, and d_bias = colsum(d_lin) where
 *      d_lin = dx * dropout_keep/(1-p) is the gradient of the Linear output that was dropped out and
 *      added to the residual before this LayerNorm (d_lin may be NULL when dropout_p == 0: d_lin == dx).
 *      `partials` is workspace of xfmr_layernorm_bwd_workspace(rows,H) bytes.
 * ---------------------------------------------------------------------------------------------- */
int xfmr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t rows, int32_t H, float eps, void* stream);
size_t xfmr_layernorm_bwd_workspace(int64_t rows, int32_t H);
int xfmr_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                       float* dx, float* d_lin, float* d_gamma, float* d_beta, float* d_bias, int64_t rows,
                       int32_t H, float dropout_p, uint64_t seed, uint32_t site, void* partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Linear layers (torch.nn.Linear at TF:models/bert/modeling_bert.py:175-177, :289, :337, :347) as MFMA
 * GEMMs with fused epilogues. x (M,K), w (N,K) [torch layout: out_features x in_features], y (M,N).
 *   XFMR_EPI_BIAS          y = x w^T + b
 *   XFMR_EPI_BIAS_GELU     pre = x w^T + b -> aux_out (M,N); y = gelu_erf(pre)        (:337-338)
 *   XFMR_EPI_BIAS_DROP_RES y = dropout(x w^T + b) + residual                          (:289-291, :347-349)
 * ---------------------------------------------------------------------------------------------- */
enum { XFMR_EPI_BIAS = 0, XFMR_EPI_BIAS_GELU = 1, XFMR_EPI_BIAS_DROP_RES = 2 };
int xfmr_linear_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int32_t N, int32_t K,
                    int32_t epilogue, const float* residual, float* aux_out, float dropout_p, uint64_t seed,
                    uint32_t site, int32_t precision, void* stream);
/* dx = dy w (+ residual_grad) or, with gelu_pre != NULL, dx = (dy w) * gelu'(gelu_pre).   dy (M,N), w (N,K), dx (M,K) */
int xfmr_linear_bwd_dx(const float* dy, const float* w, float* dx, int64_t M, int32_t N, int32_t K,
                       const float* residual_grad, const float* gelu_pre, int32_t precision, void* stream);
/* dw = dy^T x  (N,K), reduced over M with deterministic split-K slabs in `workspace`. */
size_t xfmr_linear_bwd_dw_workspace(int64_t M, int32_t N, int32_t K);
int xfmr_linear_bwd_dw(const float* dy, const float* x, float* dw, int64_t M, int32_t N, int32_t K,
                       int32_t precision, void* workspace, size_t workspace_bytes, void* stream);
/* out[n] = sum_m a[m,n] (bias gradients). workspace: xfmr_colsum_workspace(M,N) bytes. */
size_t xfmr_colsum_workspace(int64_t M, int32_t N);
int xfmr_colsum(const float* a, float* out, int64_t M, int32_t N, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K5: causal self-attention with key-padding mask, flash-style (no (L,L) score tensor in HBM).
 * Replaces TF:models/bert/modeling_bert.py:111-136 (eager_attention_forward) / SDPA and the mask built
 * at TF:masking_utils.py:76-80,168-179: key k is visible to query q iff k <= q and key_mask[b,k].
 *   qkv (B*L,3H): [q | k | v] per token, heads along H; ctx (B*L,H); lse (B,A,L) log-sum-exp of the
 *   scaled scores (saved for backward). head size H/A must be 32 or 64. A query with no visible key gets ctx = 0.
 * bwd: d_qkv (B*L,3H) from d_ctx, recomputing probabilities from qkv and lse.
 * ---------------------------------------------------------------------------------------------- */
int xfmr_attn_fwd(const float* qkv, const uint8_t* key_mask, float* ctx, float* lse, int32_t B, int32_t L,
                  int32_t A, int32_t H, float dropout_p, uint64_t seed, uint32_t site, int32_t precision,
                  void* stream);
int xfmr_attn_bwd(const float* qkv, const uint8_t* key_mask, const float* ctx, const float* lse,
                  const float* d_ctx, float* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p,
                  uint64_t seed, uint32_t site, int32_t precision, void* stream);
/* The same pair with the mask selectable: attn_mode = XFMR_ATTN_CAUSAL is exactly the pair above;
 * XFMR_ATTN_BIDIRECTIONAL drops the `k <= q` condition (key k is visible to every query iff key_mask[b,k]). */
int xfmr_attn_fwd_mode(const float* qkv, const uint8_t* key_mask, float* ctx, float* lse, int32_t B, int32_t L,
                       int32_t A, int32_t H, float dropout_p, uint64_t seed, uint32_t site, int32_t precision,
                       int32_t attn_mode, void* stream);
int xfmr_attn_bwd_mode(const float* qkv, const uint8_t* key_mask, const float* ctx, const float* lse,
                       const float* d_ctx, float* d_qkv, int32_t B, int32_t L, int32_t A, int32_t H, float dropout_p,
                       uint64_t seed, uint32_t site, int32_t precision, int32_t attn_mode, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Whole encoder (BertEmbeddings + `layers` x BertLayer), forward and backward, as one call each.
 * Replaces BertModel.forward as driven by xfmr_rec/models.py:343-345 (forward) and autograd (backward).
 * `params`/`grads`: flat buffers described above. `acts`: activation workspace of
 * xfmr_encoder_workspace_bytes(cfg) bytes, written by fwd and read by bwd (the caller keeps it alive
 * in between). tok (B*L,H) = last_hidden_state; key_mask (B*L) uint8.
 * bwd overwrites `grads` (it does not accumulate) from d_tok (B*L,H); d_tok is clobbered.
 * ---------------------------------------------------------------------------------------------- */
size_t xfmr_encoder_workspace_bytes(const xfmr_encoder_cfg* cfg);
int xfmr_encoder_fwd(const xfmr_encoder_cfg* cfg, const float* params, const int64_t* item_idx,
                     const float* table, int64_t n_rows, float* tok, uint8_t* key_mask, void* acts,
                     size_t acts_bytes, void* stream);
int xfmr_encoder_bwd(const xfmr_encoder_cfg* cfg, const float* params, float* grads, float* d_tok,
                     const uint8_t* key_mask, void* acts, size_t acts_bytes, void* stream);

/* Pooling over the token axis: sentence-transformers Pooling(pooling_mode) as wired at xfmr_rec/models.py:143-145
 * for the modes ModelConfig.pooling_mode allows (models.py:47): mean = sum_t tok*m / max(sum_t m, 1e-9);
 * max = max_t (m ? tok : -1e9); cls = tok[:,0]; lasttoken = the last token with m != 0 (zero if none).
 * Forward only: sentence_embedding is not on the training path (the loss reads token_embeddings). */
enum { XFMR_POOL_MEAN = 0, XFMR_POOL_MAX = 1, XFMR_POOL_CLS = 2, XFMR_POOL_LASTTOKEN = 3 };
int xfmr_pool(const float* tok, const uint8_t* key_mask, float* out, int32_t B, int32_t L, int32_t H, int32_t mode,
              void* stream);
int xfmr_mean_pool(const float* tok, const uint8_t* key_mask, float* out, int32_t B, int32_t L, int32_t H,
                   void* stream);

/* Row-wise L2 normalisation y = x / max(|x|, eps) and its backward: torch.nn.functional.normalize on the query
 * embeddings when ModelConfig.is_normalized (xfmr_rec/models.py:393-394, eps 1e-12) and the sentence-transformers
 * Normalize module on sentence_embedding (models.py:146-147). inv_norm (rows) is saved by fwd for bwd (may be NULL
 * in fwd when no backward follows). */
int xfmr_l2_normalize_fwd(const float* x, float* y, float* inv_norm, int64_t rows, int32_t H, float eps, void* stream);
int xfmr_l2_normalize_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t rows, int32_t H,
                          float eps, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K8-K17: fused in-batch sampled loss (all seven heads + LogitsStatistics + dL/dquery in ONE pass over
 * the logits, which are never materialised).
 * Replaces xfmr_rec/models.py:390-416 (valid-token compaction, pos/neg gathers, the (Np,1+N,H)
 * candidate tensor), xfmr_rec/losses.py:128-155 (EmbedLoss.forward: logits, target, false-negative
 * mask), :338-372, :408-543 (the seven heads), :375-405 (LogitsStatistics) and their autograd.
 *
 * Mathematically logits = [ rowdot(q, E[pos]) | Q E[neg]^T ]. Rows: positions with
 * key_mask != 0 and pos_idx != 0 (Np). Columns: the positive, then one negative per position with
 * key_mask != 0 (N), every row scoring against all of them (shared in-batch negatives).
 *
 * `mode`:
 *   XFMR_NEG_SHARED  in-batch negatives as above (reference training path).
 *   XFMR_NEG_CATALOG columns = every row of `table` (full-catalogue softmax: EmbedLoss.forward(q,
 *                    table[None].expand, target=pos_idx) with target_position=None);
 *                    neg_idx is ignored; the positive's own column is not a negative.
 * Ties: a negative that is the row's positive item has, in the reference, a logit bit-identical to the
 * positive's (same vectors through the same bmm) -- the kernel reproduces that by item id.
 * num_hard_negatives (losses.py:295-330) > 0 keeps, per row, only the k counted negatives with the largest
 * logits (cosine logits for the cosine heads): the logits are then dumped once (positions x columns fp32 in the
 * workspace: size it with the *_workspace_cfg functions), per-row thresholds come from a radix select, and the loss
 * pass weights every negative by its top-k weight (ties at the threshold share the remaining weight).
 *
 * Outputs (all device memory):
 *   losses[14]       [0..6] summed loss per head (accumulated in fp64), order XFMR_LOSS_*;
 *                    [7..13] the same divided by (Np + 1e-9): the `loss/<Class>Mean` values of trainer.py:263
 *   stats[16]        see XFMR_STAT_*; fp32. With all_heads == 0 only the counts (N_VALID, N_QUERY, NEG_DISTINCT and the
 *                    two batch densities) are defined; the logits/{pos,neg} statistics belong to the logging pass
 *   d_tok (B*L,H)    dL(train_head)/d tok, rows that are not queries are zero; may be NULL (eval)
 * workspace: xfmr_sampled_loss_workspace(B*L, H, n_rows) bytes. table_bf16: see xfmr_table_prepare (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
enum { XFMR_NEG_SHARED = 0, XFMR_NEG_CATALOG = 1 };
enum {
  XFMR_STAT_N_VALID = 0,     /* N : positions with key_mask != 0  (batch/attention_non_zero)  */
  XFMR_STAT_N_QUERY = 1,     /* Np: rows with a non-padding positive (batch/positive_non_zero) */
  XFMR_STAT_NEG_DENSITY = 2, /* logits/neg/density */
  XFMR_STAT_POS_MEAN = 3, XFMR_STAT_POS_STD = 4, XFMR_STAT_POS_MIN = 5, XFMR_STAT_POS_MAX = 6,
  XFMR_STAT_NEG_MEAN = 7, XFMR_STAT_NEG_STD = 8, XFMR_STAT_NEG_MIN = 9, XFMR_STAT_NEG_MAX = 10,
  XFMR_STAT_NEG_COUNT = 11,  /* number of (row, column) pairs counted as negatives */
  XFMR_STAT_NEG_DISTINCT = 12, /* columns the kernels walked: distinct negative items (shared mode), catalogue rows
                                  (catalogue mode), C (dense form). In-batch negatives repeat items; every per-column
                                  term is a function of the item, so it is evaluated once per distinct item and
                                  weighted by the item's multiplicity (exactly the reference's sums).             */
  XFMR_STAT_POS_DENSITY = 13,  /* batch/positive_density  = Np / (N + 1e-9)            (trainer.py:241-249) */
  XFMR_STAT_ATTN_DENSITY = 14, /* batch/attention_density = N / (positions + 1e-9)                           */
  XFMR_NUM_STATS = 16
};
typedef struct xfmr_loss_cfg {
  int32_t train_head;           /* XFMR_LOSS_*: the head whose gradient is produced               */
  int32_t all_heads;            /* 1: evaluate all seven heads + statistics (trainer.py:250-264);
                                   0: only train_head (others are returned as 0);
                                   2: (d_tok == NULL only) all heads EXCEPT train_head, which is returned as 0 --
                                      the logging half of a step whose gradient half already produced it        */
  int32_t mask_false_negatives; /* LossConfig.mask_false_negatives (losses.py:27)                   */
  int32_t mode;                 /* XFMR_NEG_*                                                       */
  int32_t precision;            /* XFMR_PREC_*                                                      */
  float scale;                  /* LossConfig.scale  (losses.py:29)                                 */
  float margin;                 /* LossConfig.margin (losses.py:30)                                 */
  int32_t num_hard_negatives;   /* LossConfig.num_hard_negatives (losses.py:28, :295-330); 0 = off    */
  /* ---- ABI 2 (zero / NULL = off) ---- */
  uint32_t flags;               /* XFMR_LOSS_* bits */
  void* profile_grad[2];        /* hipEvent_t pair recorded on `stream` right before / after the GRADIENT-pass main
                                   kernel of this call (or its only main kernel); measurement only (bench.py)      */
  void* profile_log[2];         /* ... around the values-only LOGGING pass (all seven heads + statistics,
                                   trainer.py:250-264) when this call runs one beside a gradient pass             */
  /* ---- ABI 3 ---- */
  int64_t padded_positions;     /* packed rows (xfmr_encoder_cfg.seq_offsets): the batch's PADDED position count B x L, the
                                   denominator of batch/attention_density (trainer.py:241-249 divides by the padded
                                   mask's numel); 0 = `positions`                                                  */
} xfmr_loss_cfg;
enum {
  XFMR_LOSS_DTOK_ZEROED = 1u    /* xfmr_sampled_loss_prepared: d_tok is already zero-filled (rows that are not queries
                                   must read 0) -- the caller did it, e.g. on another stream underneath the encoder
                                   forward (52 MB at T = 102 400) -- so the call skips its memset                  */
};
/* Launch-plan overrides inside `flags` (0 = the library's own plan, which depends on the number of 128-query blocks):
 * bits 8-15 the column-split count of the plan (the logging pass's), bits 16-23 the gradient pass's (clamped to the
 * former). Results do not depend on the plan beyond fp32 summation order; the parity tests walk several plans in one
 * process with these (tests/test_gpu_fullsize.py). Workspaces must be sized with the *_workspace_cfg functions. */
#define XFMR_LOSS_NSPLIT(n) (((uint32_t)(n) & 0xffu) << 8)
#define XFMR_LOSS_NSPLIT_GRAD(n) (((uint32_t)(n) & 0xffu) << 16)
#define XFMR_LOSS_NSPLIT_OF(flags) (((flags) >> 8) & 0xffu)
#define XFMR_LOSS_NSPLIT_GRAD_OF(flags) (((flags) >> 16) & 0xffu)
size_t xfmr_sampled_loss_workspace(int64_t positions, int32_t H, int64_t n_rows);            /* num_hard_negatives == 0 */
size_t xfmr_sampled_loss_workspace_cfg(const xfmr_loss_cfg* cfg, int64_t positions, int32_t H, int64_t n_rows);
int xfmr_sampled_loss(const xfmr_loss_cfg* cfg, const float* tok, const uint8_t* key_mask, const int64_t* pos_idx,
                      const int64_t* neg_idx, const float* table, const float* table_rnorm, const void* table_bf16,
                      int64_t n_rows, int64_t positions, int32_t H, float* losses, float* stats, float* d_tok, void* workspace,
                      size_t workspace_bytes, void* stream);
/* List form of the same computation, for queries that are already compacted -- the calling convention of
 * EmbedLoss.forward(query_embed (Np,H), candidate_embed) (xfmr_rec/losses.py:128-155) when the candidates are
 * the structured [positive | shared negatives] set that compute_embeds builds (xfmr_rec/models.py:398-416):
 *   query (Np,H); pos_items (Np) item id of each row's positive; neg_items (N) item ids of the shared negative
 *   columns (ignored in XFMR_NEG_CATALOG mode); d_query (Np,H) or NULL. Every row of d_query is written. */
size_t xfmr_sampled_loss_lists_workspace(int64_t n_query, int64_t n_neg, int32_t H, int64_t n_rows);
size_t xfmr_sampled_loss_lists_workspace_cfg(const xfmr_loss_cfg* cfg, int64_t n_query, int64_t n_neg, int32_t H,
                                             int64_t n_rows);
int xfmr_sampled_loss_lists(const xfmr_loss_cfg* cfg, const float* query, const int64_t* pos_items,
                            const int64_t* neg_items, int64_t n_query, int64_t n_neg, const float* table,
                            const float* table_rnorm, const void* table_bf16, int64_t n_rows, int32_t H,
                            float* losses, float* stats, float* d_query, void* workspace, size_t workspace_bytes, void* stream);
/* Dense-candidate form: EmbedLoss.forward(query_embed (N,H), candidate_embed (N,C,H), target) exactly as the
 * reference declares it (xfmr_rec/losses.py:128-155), for candidate tensors that exist in memory (C <= 8192,
 * H <= 1024, H % 4 == 0): dot / cosine logits (:179-208), target from target_position "first" / "diagonal" or an
 * explicit `target` (N) of column indices (:211-261), false-negative mask (:263-293), top-k hard negatives
 * (:295-330, ties at the k-th logit share the remaining weight), the seven heads + LogitsStatistics, and
 * d_query (N,H) = dL(train_head)/dquery (may be NULL). cfg->mode and cfg->precision are ignored (fp32 vector arithmetic).
 * losses[14] / stats[16] as for xfmr_sampled_loss, with N_VALID = C and N_QUERY = N. */
enum { XFMR_TARGET_FIRST = 0, XFMR_TARGET_DIAGONAL = 1, XFMR_TARGET_EXPLICIT = 2 };
size_t xfmr_dense_loss_workspace(int64_t N, int32_t C, int32_t H);
int xfmr_dense_loss(const xfmr_loss_cfg* cfg, const float* query, const float* cand, const int64_t* target,
                    int32_t target_mode, int64_t N, int32_t C, int32_t H, float* losses, float* stats, float* d_query,
                    void* workspace, size_t workspace_bytes, void* stream);
/* ... and with d_cand (N,C,H) = dL(train_head)/dcandidate_embed (may be NULL): EmbedLoss.forward is differentiable in
 * both arguments (losses.py:128-155; cosine heads through F.normalize of the candidates, losses.py:196-208). */
int xfmr_dense_loss_grads(const xfmr_loss_cfg* cfg, const float* query, const float* cand, const int64_t* target,
                          int32_t target_mode, int64_t N, int32_t C, int32_t H, float* losses, float* stats,
                          float* d_query, float* d_cand, void* workspace, size_t workspace_bytes, void* stream);
/* xfmr_sampled_loss in two halves. _prepare: everything that depends only on the key mask and the index tensors (the
 * compacted query list, multiplicities and the distinct-item list of the shared negatives) -- 7 small launches that a
 * caller can enqueue as soon as the key mask exists (xfmr_encoder_cfg.embed_event), on another stream, instead of between
 * the encoder forward and the loss kernels. _prepared: the rest, on a workspace _prepare filled for the same cfg, key
 * mask, index tensors and sizes. xfmr_sampled_loss == _prepare followed by _prepared on one stream. */
int xfmr_sampled_loss_prepare(const xfmr_loss_cfg* cfg, const uint8_t* key_mask, const int64_t* pos_idx,
                              const int64_t* neg_idx, const float* table_rnorm, int64_t n_rows, int64_t positions,
                              int32_t H, void* workspace, size_t workspace_bytes, void* stream);
int xfmr_sampled_loss_prepared(const xfmr_loss_cfg* cfg, const float* tok, const uint8_t* key_mask, const int64_t* pos_idx,
                               const int64_t* neg_idx, const float* table, const float* table_rnorm, const void* table_bf16,
                               int64_t n_rows, int64_t positions, int32_t H, float* losses, float* stats, float* d_tok,
                               void* workspace, size_t workspace_bytes, void* stream);
/* table_rnorm[r] = 1 / max(||table[r]||, 1e-8): per-item inverse norms for the cosine heads
 * (torch cosine_similarity, losses.py:206-208); computed once because the table is frozen. */
int xfmr_table_rnorm(const float* table, float* table_rnorm, int64_t n_rows, int32_t H, void* stream);
/* Same, plus (optionally) table_bf16 (n_rows,H): a bf16 copy of the frozen table. When it is passed to
 * xfmr_sampled_loss[_lists] with XFMR_PREC_BF16, negatives are gathered from it by LDS-DMA (no staging
 * registers); with NULL the fp32 table is converted on the fly. Either output may be NULL. */
int xfmr_table_prepare(const float* table, float* table_rnorm, void* table_bf16, int64_t n_rows, int32_t H,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side sequence sampler: SeqDataset.__getitem__ + collate of the reference
 * (xfmr_rec/data.py:669-805) for a whole batch in one launch.
 *   items / labels (nnz): all users' histories back to back (item index 1..n_items, positive-label flag), as
 *   process_events leaves them (data.py:590-656); offsets (R+1): row r = [offsets[r], offsets[r+1]);
 *   rows (batch): the dataset rows of this batch. Per row: positions 0..n-2, at most max_seq_length of them uniformly
 *   without replacement, sorted (data.py:669-688); for each a positive drawn uniformly from the later
 *   positive-labelled items, within pos_lookahead if > 0, else 0 (data.py:690-722); as many negatives, uniform over
 *   the catalogue minus the row's history, without replacement while possible (data.py:724-747); everything
 *   right-padded with 0 to `width` (pad_sequence, data.py:789-805; width = the batch's longest row, <= max_seq_length).
 * Outputs hist_out / pos_out / neg_out: (batch, width) int64 -- exactly the three index tensors of SeqBatch.
 * max_history >= the longest history among `rows` (<= 8192). Random stream: counter-based hash of
 * (seed, row, purpose, counter) -- reproducible, but not numpy's generator: parity is distributional.
 * ---------------------------------------------------------------------------------------------- */
size_t xfmr_seq_sample_workspace(int32_t batch, int64_t n_items);
int xfmr_seq_sample(const int64_t* items, const uint8_t* labels, const int64_t* offsets, const int64_t* rows,
                    int32_t batch, int32_t width, int32_t max_seq_length, int32_t pos_lookahead, int64_t n_items,
                    int32_t max_history, uint64_t seed, int64_t* hist_out, int64_t* pos_out, int64_t* neg_out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Exact top-k retrieval and ranking metrics for validation: replaces the LanceDB
 * IVF_HNSW_PQ search with the history prefiltered out (xfmr_rec/index.py:214-255, called from trainer.py:186-211,
 * 305-314) by an exact scan -- the ANN's limit of full probing -- and torchmetrics' seven retrieval metrics on the
 * synthesised score vector of xfmr_rec/metrics.py:17-79.
 *   xfmr_topk: query (n_query,H) sentence embeddings; table (n_rows,H) item embeddings, row 0 = padding (never
 *   returned); exclude / exclude_offsets: CSR of item indices to leave out per query (NULL, NULL: none);
 *   metric XFMR_METRIC_* (index.py:47, default cosine); out_idx (n_query,k) item indices best first, -1 where fewer
 *   than k items remain; out_score = 1 - distance (index.py:248-251): cosine similarity, dot, or 1 - |q - e|^2.
 *   Ties are broken by the lower item index. k <= 1024, H <= 1024.
 *   xfmr_retrieval_metrics: rec_idx (n_query,k) as above; targets / target_offsets: CSR of each query's positive
 *   item indices; out (n_query,7) = nDCG, MAP, AUROC, precision, recall, hit rate, MRR at top_k (XFMR_RM_*);
 *   valid[q] = 0 when the query has no target (the reference reports nothing for it, metrics.py:58-59).
 * ---------------------------------------------------------------------------------------------- */
enum { XFMR_METRIC_COSINE = 0, XFMR_METRIC_DOT = 1, XFMR_METRIC_L2 = 2 };
enum { XFMR_RM_NDCG = 0, XFMR_RM_MAP = 1, XFMR_RM_AUROC = 2, XFMR_RM_PRECISION = 3, XFMR_RM_RECALL = 4,
       XFMR_RM_HIT_RATE = 5, XFMR_RM_MRR = 6, XFMR_NUM_RM = 7 };
size_t xfmr_topk_workspace(int64_t n_query, int64_t n_rows);
int xfmr_topk(const float* query, const float* table, const float* table_rnorm, int64_t n_rows, int64_t n_query,
              int32_t H, const int64_t* exclude, const int64_t* exclude_offsets, int32_t k, int32_t metric,
              int64_t* out_idx, float* out_score, void* workspace, size_t workspace_bytes, void* stream);
int xfmr_retrieval_metrics(const int64_t* rec_idx, const int64_t* targets, const int64_t* target_offsets, int32_t n_query,
                           int32_t k, int32_t top_k, float* out, uint8_t* valid, void* stream);

/* ------------------------------------------------------------------------------------------------
 * K18: AdamW over the flat buffer (torch.optim.AdamW as configured at xfmr_rec/trainer.py:327-332:
 * decoupled weight decay, bias-corrected moments, eps outside the sqrt). `step` is 1-based.
 * grad_scale multiplies the gradient first (1/world_size after a SUM all-reduce).
 * ---------------------------------------------------------------------------------------------- */
int xfmr_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
               float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
               void* stream);

/* The same update with the step count read from device memory: step = *step_device + step_offset (1-based, as above).
 * With xfmr_step_advance(step_device) as the last launch of a step, a captured step replays with the right bias
 * corrections and (xfmr_encoder_cfg.step_device = the same counter) a new dropout mask. */
int xfmr_adamw_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, const uint32_t* step_device,
                   int32_t step_offset, float grad_scale, void* stream);
int xfmr_step_advance(uint32_t* step_device, void* stream); /* *step_device += 1 (one thread) */

/* ------------------------------------------------------------------------------------------------
 * K19: the data-parallel exchange (SURVEY section 8b `allreduce_flat`, 8e): what torch DDP does for the reference
 * (config.yaml:5-6,35 -- Lightning `strategy: auto`): g <- SUM over ranks of the flat gradient buffer, in place, fp32, on
 * the caller's stream; the 1 / world factor is xfmr_adamw's grad_scale. RCCL does the transport (rings over xGMI); it is
 * loaded at RUN time from the process (dlopen of librccl.so.1: the copy already mapped -- torch's -- or the ROCm one), so
 * nothing here links against it and a single-GPU user never loads it.
 *   rank 0:        xfmr_comm_unique_id(id)            ... ship the 128 bytes to every rank by any channel ...
 *   every rank:    xfmr_comm_create(&comm, id, world, rank)   (collective; binds the CURRENT device)
 *   every step:    xfmr_allreduce_flat(comm, grads, n, stream)
 *   at the end:    xfmr_comm_destroy(comm)
 * The communicator is the one object these calls create; the caller owns it. XFMR_ECOMM: RCCL missing or an RCCL error
 * (xfmr_comm_last_error(): RCCL's own text for the last failure of this thread).
 * ---------------------------------------------------------------------------------------------- */
#define XFMR_COMM_ID_BYTES 128
int xfmr_comm_unique_id(unsigned char id[XFMR_COMM_ID_BYTES]);
int xfmr_comm_create(void** comm, const unsigned char id[XFMR_COMM_ID_BYTES], int32_t world, int32_t rank);
int xfmr_comm_destroy(void* comm);
int xfmr_allreduce_flat(void* comm, float* grads, int64_t n, void* stream);
const char* xfmr_comm_last_error(void);

/* Elementwise helpers used by the autograd wrappers. */
int xfmr_scale_by_device_scalar(float* x, int64_t n, const float* scalar, void* stream);

/* Self-test of the MFMA operand / accumulator lane maps this library relies on (exact integers, asymmetric
 * operands, both precisions) and of the swizzled LDS-DMA gather image with its row and transposed
 * (ds_read_b64_tr_b16) fragment reads. `out`: >= 16 KiB + 16 B of device memory; out[0] = mismatches of the
 * plain maps in total, out[1] = bf16, out[2] = f32, out[3] = swizzled-image path. */
int xfmr_selftest_mfma(int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XFMR_HIP_H */
