// Host-side runtime objects of the library: streams, events and the host -> HBM hand-over of a collated batch. No kernels.
// Everything here is owned by the caller (create / destroy pairs); none of it synchronises the host.
#include "internal.h"

extern "C" {

int xfmr_low_priority_stream_create(void** stream) {
  if (!stream) return XFMR_EINVAL;
  int lo = 0, hi = 0;
  hipStream_t s = nullptr;
  if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess ||
      hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo) != hipSuccess)
    return XFMR_EHIP;
  *stream = s;
  return XFMR_OK;
}
int xfmr_stream_create(void** stream) {
  if (!stream) return XFMR_EINVAL;
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return XFMR_EHIP;
  *stream = s;
  return XFMR_OK;
}
int xfmr_stream_destroy(void* stream) {
  if (!stream) return XFMR_EINVAL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}

int xfmr_event_create(void** event, int32_t timing) {
  if (!event) return XFMR_EINVAL;
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming) != hipSuccess) return XFMR_EHIP;
  *event = e;
  return XFMR_OK;
}
int xfmr_event_destroy(void* event) {
  if (!event) return XFMR_EINVAL;
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}
int xfmr_event_record(void* event, void* stream) {
  if (!event) return XFMR_EINVAL;
  return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}
int xfmr_stream_wait_event(void* stream, void* event) {
  if (!event) return XFMR_EINVAL;
  return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}
int xfmr_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!start || !stop || !ms) return XFMR_EINVAL;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}

int xfmr_event_synchronize(void* event) {
  if (!event) return XFMR_EINVAL;
  return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? XFMR_OK : XFMR_EHIP;
}
int xfmr_event_query(void* event) {  // 1: complete, 0: not yet
  if (!event) return XFMR_EINVAL;
  const hipError_t e = hipEventQuery((hipEvent_t)event);
  return e == hipSuccess ? 1 : e == hipErrorNotReady ? 0 : XFMR_EHIP;
}

// (Measured, scripts/probe/h2d_probe.py: a hipStreamWaitEvent on the copy stream in front of the hipMemcpyAsync makes the
// COPY CALL block the host until that event has completed -- the runtime resolves a copy's dependencies on the host --
// 890 us per step at the bench shape, and the host loses its run-ahead. The write-after-read guard of a slot is therefore
// a host-side wait on the slot's "free" event, which only ever blocks when the host is a whole ring ahead of the GPU.)
int xfmr_batch_upload(void* dst_device, const void* src_pinned, size_t bytes, void* copy_stream, void* slot_free_event,
                      void* ready_event) {
  if (!dst_device || !src_pinned || !bytes || !ready_event) return XFMR_EINVAL;
  hipStream_t cs = (hipStream_t)copy_stream;
  if (slot_free_event && hipEventSynchronize((hipEvent_t)slot_free_event) != hipSuccess) return XFMR_EHIP;
  if (hipMemcpyAsync(dst_device, src_pinned, bytes, hipMemcpyHostToDevice, cs) != hipSuccess) return XFMR_EHIP;
  if (hipEventRecord((hipEvent_t)ready_event, cs) != hipSuccess) return XFMR_EHIP;
  return XFMR_OK;
}

}  // extern "C"
