// shadow.hpp -- resident F16 images of quantised WEIGHT tensors for the prefill GEMM.
//
// MI355X has 288 GB of HBM per GPU: a Q4_K_M Qwen3-8B is 4.7 GB and its F16 image 15 GB, so both stay resident.  Decode
// (bandwidth-bound) streams the quantised blocks; prefill (compute-bound) feeds the matrix cores from the F16 image instead of
// de-quantising every weight on every ubatch (what the reference does per call: ggml-cuda.cu:1250-1268, to_fp16_cuda into a pool
// buffer).  An image is built the first time a weight is used by a GEMM, only for tensors whose buffer carries
// GGML_BACKEND_BUFFER_USAGE_WEIGHTS (what libllama sets on model buffers, src/llama-model.cpp), and is dropped when the
// tensor's bytes are rewritten through the buffer interface (set_tensor / memset / cpy / clear / free).
#pragma once
#include "common.hpp"

namespace mi {

// The image of rows [M x K] at `src` (row stride src_rs) for use on stream `st`, or null when there is none (yet).  An image is filled by a
// kernel on its creator's stream: until that kernel is known to have finished, other streams are made to wait on the image's event
// (hipStreamWaitEvent) -- or, while `capturing` (no cross-stream waits inside a stream capture), get null and take the per-call path.
const uint16_t * shadow_find(int device, const void * src, int type, int64_t K, int64_t M, size_t src_rs, hipStream_t st, bool capturing);
// Find-or-create under ONE lock (two contexts sharing a model never register the same image twice).  Returns null when shadows are
// disabled, memory is short or another thread is still filling this image; *created tells the caller to launch the fill kernel on `st`
// and then call shadow_mark_ready(image, st).
uint16_t *       shadow_get_or_create(int device, const void * src, size_t src_bytes, int type, int64_t K, int64_t M, size_t src_rs, hipStream_t st, bool capturing, bool * created);
void             shadow_mark_ready(const void * image, hipStream_t st);
// drop every image of the device (an allocation failed: the images are the first thing to give back); returns the bytes freed
// Images are shared by every backend context of a device (the omni pipeline runs LLM / TTS / Token2Wav on separate threads): a context that looks images up and
// enqueues launches reading them holds the device's image table as a READER for that whole submission (graph_compute does: shadow_reader); shadow_drop_all takes it
// exclusively, so it frees only when no submission is between "pointer looked up" and "launch enqueued" -- hipFree then waits for the enqueued work itself.
struct shadow_reader { int device; bool held; explicit shadow_reader(int device); ~shadow_reader(); void unlock(); void lock(); };
size_t           shadow_drop_all(int device, shadow_reader * mine = nullptr);      // mine: the caller's own reader hold (released around the exclusive section)
size_t           shadow_bytes(int device);
// bytes [p, p+n) of device memory are about to change: drop the overlapping images
void             shadow_invalidate(int device, const void * p, size_t n);
// bumped whenever an image is dropped: captured hipGraphs that baked an image pointer in are stale after that
uint64_t         shadow_generation();
void             shadow_set_enabled(bool on);
bool             shadow_enabled();
double           shadow_stat(const char * name);      // "shadow_bytes", "shadow_tensors"

} // namespace mi
