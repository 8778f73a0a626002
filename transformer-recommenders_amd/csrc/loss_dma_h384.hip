// LDS-DMA bf16 main kernel of the fused sampled loss, H = 384 (the reference's default d_model): six gradient-pass instantiations (one per head
// with a negative term) + the logging pass. One translation unit per hidden size to keep build time down.
#include "loss_common.h"
#include "loss_dma.inc"

int xf_launch_loss_dma_384(const LossArgs& a, const void* table_bf16, int head, dim3 grid, hipStream_t st) {
  const __bf16* tbf = (const __bf16*)table_bf16;
  dim3 block(256);
  if (head == XFMR_LOSS_INFONCE && !a.mask_fn && !a.pin_part) grid.z = 3;  // three dQ column parts (loss_dma.inc)
  switch (head) {
    // (the masked fast-path epilogue of the logging pass is for H <= 128 only: at one wave per SIMD it measured 26 %
    //  SLOWER than the general epilogue -- 2.21 against 1.75 ms at BASELINE config 5)
    case -1: hipLaunchKernelGGL((loss_main_dma_kernel<384, -1>), grid, block, 0, st, a, tbf); break;
    case -2: hipLaunchKernelGGL((loss_main_dma_kernel<384, -2>), grid, block, 0, st, a, tbf); break;
    case XFMR_LOSS_ALIGNMENT_CONTRASTIVE:  // masking on + in-batch negatives: the lean cosine epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED)
        hipLaunchKernelGGL((loss_main_dma_kernel<384, HEAD_CCL_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_ALIGNMENT_CONTRASTIVE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_CONTRASTIVE:  // masking on + in-batch negatives: the lean cosine epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED)
        hipLaunchKernelGGL((loss_main_dma_kernel<384, HEAD_CONTR_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_CONTRASTIVE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_INFONCE:
      if (a.mask_fn) hipLaunchKernelGGL((loss_main_dma_kernel<384, HEAD_INFONCE_MASKED>), grid, block, 0, st, a, tbf);
      else if (a.pin_part) hipLaunchKernelGGL((loss_main_dma_kernel<384, HEAD_INFONCE_PINNED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_INFONCE>), grid, block, 0, st, a, tbf);
      break;
    case XFMR_LOSS_NCE:
      hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_NCE>), grid, block, 0, st, a, tbf); break;
    case XFMR_LOSS_PAIRWISE_HINGE:
      hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_PAIRWISE_HINGE>), grid, block, 0, st, a, tbf); break;
    case XFMR_LOSS_PAIRWISE_LOGISTIC:  // BPR: masking on + in-batch negatives (the reference's training form) take the lean epilogue
      if (a.mask_fn && a.mode == XFMR_NEG_SHARED && !a.tau)
        hipLaunchKernelGGL((loss_main_dma_kernel<384, HEAD_BPR_MASKED>), grid, block, 0, st, a, tbf);
      else hipLaunchKernelGGL((loss_main_dma_kernel<384, XFMR_LOSS_PAIRWISE_LOGISTIC>), grid, block, 0, st, a, tbf);
      break;
    default: return XFMR_EINVAL;
  }
  XF_LAUNCH_CHECK();
  return XFMR_OK;
}
