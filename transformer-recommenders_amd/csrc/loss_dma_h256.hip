// LDS-DMA bf16 main kernel of the fused sampled loss, H = 256: six gradient-pass instantiations (one per head
// with a negative term) + the logging pass. One translation unit per hidden size to keep build time down.
#include <stdlib.h>

#include "loss_common.h"
#include "loss_dma.inc"

static bool logm256() {
  static const bool on = [] { const char* e = getenv("XFMR_LOSS_LOGM256"); return !(e && *e == '0'); }();
  return on;
}

// dQ column parts of the gradient pass this launcher would start for `head` (loss.hip plans its splits with it)
int xf_loss_dma_256_hparts(const LossArgs& a, int head) {
  if (head == XFMR_LOSS_INFONCE && !a.mask_fn && !a.pin_part) return 2;  // the online-maximum InfoNCE (loss_dma.inc)
  if (!XFL_H256_LEAN_HALVES || !a.mask_fn) return 1;
  if (head == XFMR_LOSS_INFONCE) return 2;
  if (a.mode != XFMR_NEG_SHARED) return 1;
  if (head == XFMR_LOSS_ALIGNMENT_CONTRASTIVE || head == XFMR_LOSS_CONTRASTIVE) return 2;
  return (head == XFMR_LOSS_PAIRWISE_LOGISTIC && !a.tau) ? 2 : 1;
}

int xf_launch_loss_dma_256(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st) {
  const __bf16* tbf = (const __bf16*)table_bf16;
  dim3 block(256);
  if (head >= 0) grid.z = (unsigned)xf_loss_dma_256_hparts(a, head);  // dQ column halves (loss_dma.inc)
  switch (head) {
    // logging pass: masking on + in-batch negatives take the fast epilogue (loss_epilogue_logging_masked), as at H = 128.
    // (Round 2 measured it 26 % SLOWER than the general epilogue at ONE wave per SIMD -- 2.21 against 1.75 ms at BASELINE
    // config 5; the values-only kernels now run two waves per SIMD. XFMR_LOSS_LOGM256=0 keeps the general epilogue: A/B.)
    case -1:
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED && logm256())
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_LOG_MASKED_LSE>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, -1>), grid, block, 0, st, a, tbf);
      break;
    case -2:
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED && logm256())
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_LOG_MASKED>), grid, block, 0, st, a, tbf);
      else if (!a.mask_fn && a.mode == XFMR_NEG_CATALOG && logm256())  // full-catalogue softmax (config 4)
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_LOG_UNMASKED_CATALOG>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, -2>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_ALIGNMENT_CONTRASTIVE:  // masking on + in-batch negatives: the lean cosine epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED)
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_CCL_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_ALIGNMENT_CONTRASTIVE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_CONTRASTIVE:  // masking on + in-batch negatives: the lean cosine epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED)
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_CONTR_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_CONTRASTIVE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_INFONCE:
      if (a.mask_fn) hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_INFONCE_MASKED>), grid, block, 0, st, a, tbf);
      else if (a.pin_part) hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_INFONCE_PINNED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_INFONCE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_NCE:
      hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_NCE>), grid, block, 0, st, a, tbf); break;
    case XFMR_LOSS_PAIRWISE_HINGE:
      hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_PAIRWISE_HINGE>), grid, block, 0, st, a, tbf); break;
    case XFMR_LOSS_PAIRWISE_LOGISTIC:  // BPR: masking on + in-batch negatives (the reference's training form) take the lean epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED && !a.tau)
        hipLaunchKernelGGL((loss_main_dma_kernel<256, HEAD_BPR_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<256, XFMR_LOSS_PAIRWISE_LOGISTIC>), grid, block, 0, st, a, tbf);
      break;
    default: return XFMR_EINVAL;
  }
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
