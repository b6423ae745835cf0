/* ggml-mi355x.h -- public C ABI of libggml-mi355x.so, the MI355X (gfx950 / CDNA4) ggml backend.
 *
 * The library is a drop-in ggml backend plug-in.  The reference binds it exactly like any other
 * dynamically loaded backend:
 *
 *   loader   : ggml_backend_load(path) / $GGML_BACKEND_PATH        (reference ggml/src/ggml-backend-reg.cpp:603-607)
 *   symbols  : ggml_backend_init  (required)                       (ggml-backend-reg.cpp:265-285, typedef ggml-backend-impl.h:214)
 *              ggml_backend_score (optional, 0 = unsupported here) (ggml-backend-reg.cpp:257-263, typedef ggml-backend-impl.h:217)
 *   objects  : ggml_backend_reg / _device / _buffer_type / _buffer / ggml_backend and their *_i vtables,
 *              GGML_BACKEND_API_VERSION 2                          (ggml/src/ggml-backend-impl.h:11-210)
 *
 * Everything crossing the boundary is a plain C struct of function pointers, plain pointers and sizes.
 * No C++ or torch types appear in any signature.  The struct layouts are declared (restated, not copied) in
 * llama.cpp-omni_amd/csrc/ggml_abi.h and verified against the reference headers by tests/test_abi.py.
 */
#ifndef GGML_MI355X_H
#define GGML_MI355X_H

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_API __attribute__((visibility("default")))
#define GGML_MI355X_NAME "MI355X"          /* registry name; devices are "MI355X0", "MI355X1", ... */

#include <stddef.h>

struct ggml_backend_reg;
struct ggml_backend_buffer;
struct ggml_backend;
struct ggml_tensor;

/* replaces: the `ggml_backend_init` entry a backend module exports (GGML_BACKEND_DL_IMPL,
 * reference ggml/src/ggml-backend-impl.h:220-233).  Returns the statically owned registry object
 * (api_version == 2); never freed. */
GGML_MI355X_API struct ggml_backend_reg * ggml_backend_init(void);

/* replaces: `ggml_backend_score` (GGML_BACKEND_DL_SCORE_IMPL, ggml-backend-impl.h:234-251).
 * 100 when at least one gfx950 device is visible to the HIP runtime, 0 otherwise. */
GGML_MI355X_API int ggml_backend_score(void);

/* same object as ggml_backend_init(), for hosts that link the library directly
 * (mirrors ggml_backend_cuda_reg(), reference ggml/include/ggml-cuda.h:40). */
GGML_MI355X_API struct ggml_backend_reg * ggml_backend_mi355x_reg(void);

/* ---- extensions, also reachable through reg->iface.get_proc_address(reg, "<name>") -------------------- */

/* timing events on a backend's stream (hipEvent with timing enabled; ggml's own events carry no clock). */
GGML_MI355X_API void * mi355x_timed_event_new(void);
GGML_MI355X_API void   mi355x_timed_event_record(void * ev, struct ggml_backend * backend);
GGML_MI355X_API float  mi355x_timed_event_elapsed_ms(void * start, void * stop);      /* synchronises on `stop` */
GGML_MI355X_API void   mi355x_timed_event_free(void * ev);

/* runtime options: "graphs" (0/1 hipGraph replay of repeated cgraphs), "fusion" (0/1 node fusion),
 * "profile" (0/1 per-kernel-class event timing, disables graphs), "f16_shadow" (0/1 resident F16 images of quantised weights for
 * the prefill GEMM, process-wide), "norm_in_kernel" (0/1 RMS_NORM+MUL built inside the consuming decode mat-vec launches; default 0),
 * "fattn_gqa" (0/1 matrix-core kernel for few-token FLASH_ATTN_EXT; 0 = streaming kernel for every such shape; default 1, process-wide),
 * "fattn_dma" (-1 default / 0 / 1: the LDS-DMA ring form of the prefill attention kernel for head size 128 on large grids, process-wide),
 * "prefill_q8k" (0 default / 1: the prefill GEMMs of K-quant weights take the Q8_K-QUANTISED activations -- the reference CPU backend's vec_dot_type
 * arithmetic, ggml-cpu.c:1245-1268 -- instead of plain f16 rows; process-wide), "mmq_tile" (0 default / 1: Q4_K weights x more than 64 columns on the int8
 * matrix cores from the blocks, csrc/kernels/mmq_tile.hip: the oracle's exact integer sums, slower than the F16-image GEMM; process-wide),
 * "mv1", "mv2", "fattn_one", "kq_staging", "batch_uploads" (cross-check switches of the decode kernels),
 * "reset_stats".  Returns 0 on success, -1 for an unknown key.
 * Environment switches read once per process (measurement / cross-check only): MI355X_GRAPHS=0, MI355X_NO_CONV_FUSE, MI355X_NO_CONCAT_TAIL, MI355X_NO_ATTN_F32,
 * MI355X_NO_GEMM_F32_T16, MI355X_NO_NORM_FUSE, MI355X_NO_GATE_NORM, MI355X_NO_EW_CHAIN, MI355X_NO_CONT_SINK (each turns one graph matcher / kernel choice off);
 * MI355X_LOG_STATS, MI355X_GRAPH_GPU_TIME, MI355X_LAUNCH_LOG=file, MI355X_GRAPH_SLICE="nodes:lo:hi,..." (instrumentation, see tools/t2w_slices.sh). */
GGML_MI355X_API int    mi355x_set_option(struct ggml_backend * backend, const char * key, long value);
/* counters: "graph_replays", "graph_captures", "eager_graphs", "kernels_last_graph", "mmq_tile_launches",
 * "shadow_bytes", "shadow_tensors", "prof_mmv_q4k_us", "prof_mmv_q4k_n", "prof_mmv_q4k_bytes", ....
 * Returns -1 if unknown. */
GGML_MI355X_API double mi355x_get_stat(struct ggml_backend * backend, const char * key);

/* test hook: run the on-device activation quantiser the MUL_MAT path uses (kind 0: Q8_K image, 1: Q8_0 image,
 * layouts in csrc/common.hpp) on `nrows` host rows of K floats and return the images to host memory, so the integer
 * stage can be compared bit-for-bit with the reference's quantize_row_q8_K / quantize_row_q8_0.  Returns bytes per image. */
GGML_MI355X_API long   mi355x_debug_quantize(struct ggml_backend * backend, int kind, const float * host_x, long K, long nrows, void * host_images);

/* ---- omni pipeline: module pinning and the LLM -> TTS hidden-state hand-off (BASELINE configs[3], [4]) ----------------
 *
 * replaces: the host round trip of `LLMOut::hidden_states` (std::vector<float> filled from llama_get_embeddings, reference
 * tools/omni/omni.cpp:256-270; consumed by prefill_with_emb_tts :2081 and the projector graph :1187-1258).  With the LLM and the TTS
 * decoder on different MI355X the rows go GPU to GPU: RCCL ncclSend / ncclRecv over one xGMI link, stream-ordered behind the LLM graph on
 * `src_backend` and in front of whatever `dst_backend` runs next; no host synchronisation.  `src` / `dst` are device pointers inside
 * buffers of the respective backend (e.g. tensor->data).  Returns 1 (RCCL), 2 (peer-copy fallback: librccl absent or
 * MI355X_HANDOFF=peer), 0 (nbytes == 0), < 0 on error.  mi355x_handoff_tensor moves a dense f32 / f16 tensor into one of equal size. */
GGML_MI355X_API int    mi355x_handoff_init(void);                      /* optional: build the communicators now; returns ranks (0 = fallback) */
GGML_MI355X_API int    mi355x_handoff(struct ggml_backend * src_backend, const void * src, struct ggml_backend * dst_backend, void * dst, size_t nbytes);
GGML_MI355X_API int    mi355x_handoff_tensor(struct ggml_backend * src_backend, const struct ggml_tensor * src, struct ggml_backend * dst_backend, struct ggml_tensor * dst);
GGML_MI355X_API long   mi355x_handoff_count(int kind);                 /* hand-offs done so far through RCCL (1) / peer copies (2) */
GGML_MI355X_API void   mi355x_handoff_shutdown(void);
/* replaces: nothing in the reference (it loads every module on the default device); the device ordinal ("MI355X<i>") a module of the
 * omni pipeline is pinned to: "vpm", "apm", "llm", "tts", "t2w", "vocoder" -> 0..5 modulo the visible devices, or MI355X_MODULE_MAP. */
GGML_MI355X_API int    mi355x_module_device(const char * module);

/* standalone harness only (no libggml-base in the process): what ggml_backend_buffer_free does
 * (reference ggml/src/ggml-backend.cpp:108-117): iface.free_buffer, then delete the object. */
GGML_MI355X_API void   mi355x_host_buffer_free(struct ggml_backend_buffer * buffer);

#ifdef __cplusplus
}
#endif
#endif /* GGML_MI355X_H */
