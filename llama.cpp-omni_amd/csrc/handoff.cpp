// handoff.cpp -- the one exchange step of the omni pipeline, on the device: LLM -> TTS hidden states (north star).
//
// In the reference both ends are host memory: `LLMOut::hidden_states` is a std::vector<float> filled from llama_get_embeddings
// (tools/omni/omni.cpp:256-270), handed to the TTS thread through a queue and uploaded again by prefill_with_emb_tts (omni.cpp:2081) /
// the projector graph (omni.cpp:1187-1258): device -> host -> thread queue -> host -> device, [<= 26 tokens x 4096] f32 per chunk.
// With the modules pinned to different MI355X of one node (mi355x_module_device) the same rows can go GPU to GPU over xGMI:
//
//     mi355x_handoff(llm_backend, src, tts_backend, dst, nbytes)
//         = ncclGroupStart; ncclSend(src -> rank(tts)) on the LLM backend's stream; ncclRecv(dst <- rank(llm)) on the TTS backend's stream;
//           ncclGroupEnd                                   (RCCL point-to-point: one xGMI link, no staging through the host)
//
// Stream-ordered on both sides: the send is queued behind the LLM graph that produced `src`, the TTS graph submitted afterwards on
// `tts_backend` is queued behind the receive.  One process drives all devices (omni runs its modules as threads of one process,
// tools/omni/omni.h:194-196), so the communicators come from ncclCommInitAll.  RCCL is loaded with dlopen on first use: a host that never
// hands off never maps it.  Without RCCL (or with it disabled by MI355X_HANDOFF=peer) the copy falls back to hipMemcpyPeerAsync + an event.
#include "graph.hpp"
#include "../../include/ggml-mi355x.h"
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <string>
#include <vector>

namespace mi {

typedef void * nccl_comm_t;
struct rccl_api {
    void * handle = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
};
static std::mutex               g_ho_mu;
static rccl_api                 g_rccl;
static std::vector<nccl_comm_t> g_comms;          // one per visible device, rank == device ordinal
static std::vector<int>         g_comm_dev;
static int                      g_ho_state = 0;   // 0: not tried, 1: RCCL ready, -1: unavailable (peer-copy fallback)
static long                     g_ho_rccl_calls = 0, g_ho_peer_calls = 0;

static bool load_rccl() {
    const char * names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char * n : names) { g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl.handle) break; }
    if (!g_rccl.handle) return false;
#define MI_SYM(field, sym) g_rccl.field = (decltype(g_rccl.field)) dlsym(g_rccl.handle, sym); if (!g_rccl.field) return false
    MI_SYM(CommInitAll, "ncclCommInitAll"); MI_SYM(CommDestroy, "ncclCommDestroy"); MI_SYM(GroupStart, "ncclGroupStart"); MI_SYM(GroupEnd, "ncclGroupEnd");
    MI_SYM(Send, "ncclSend"); MI_SYM(Recv, "ncclRecv"); MI_SYM(GetErrorString, "ncclGetErrorString");
#undef MI_SYM
    return true;
}

static int handoff_init_locked() {
    if (g_ho_state != 0) return g_ho_state;
    const char * mode = getenv("MI355X_HANDOFF");
    if (mode && !strcmp(mode, "peer")) { g_ho_state = -1; return g_ho_state; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void) hipGetLastError(); g_ho_state = -1; return g_ho_state; }
    if (!load_rccl()) { log_msg(GGML_LOG_LEVEL_WARN, "[mi355x] hand-off: librccl not found, using peer copies\n"); g_ho_state = -1; return g_ho_state; }
    g_comm_dev.resize(n); g_comms.assign(n, nullptr);
    for (int i = 0; i < n; ++i) g_comm_dev[i] = i;
    int prev = 0; (void) hipGetDevice(&prev);
    const int rc = g_rccl.CommInitAll(g_comms.data(), n, g_comm_dev.data());
    (void) hipSetDevice(prev);
    if (rc != 0) {
        log_msg(GGML_LOG_LEVEL_WARN, "[mi355x] hand-off: ncclCommInitAll failed (%s), using peer copies\n", g_rccl.GetErrorString(rc));
        g_comms.clear(); g_ho_state = -1; return g_ho_state;
    }
    g_ho_state = 1;
    return g_ho_state;
}

} // namespace mi

extern "C" {

int mi355x_handoff_init(void) {
    std::lock_guard<std::mutex> lk(mi::g_ho_mu);
    return mi::handoff_init_locked() == 1 ? (int) mi::g_comms.size() : 0;
}

// Neither backend may be submitting work from another thread while the hand-off enqueues on its stream (one ggml_backend = one stream, driven
// by one thread at a time: the omni pipeline calls this from the producer's thread after its graph_compute and before the consumer's).
int mi355x_handoff(struct ggml_backend * src_backend, const void * src, struct ggml_backend * dst_backend, void * dst, size_t nbytes) {
    if (!src_backend || !dst_backend || (!src && nbytes) || (!dst && nbytes)) return -1;
    if (!mi::backend_is_mi355x(src_backend) || !mi::backend_is_mi355x(dst_backend)) return -1;     // (a CPU backend's context is not a backend_ctx)
    if (nbytes == 0) return 0;
    mi::backend_ctx * cs = (mi::backend_ctx *) src_backend->context; mi::backend_ctx * cd = (mi::backend_ctx *) dst_backend->context;
    std::lock_guard<std::mutex> lk(mi::g_ho_mu);                // (RCCL group calls of one process are not re-entrant across threads)
    int dev_before = 0;
    HIP_CHECK(hipGetDevice(&dev_before));                       // the caller's current device is restored on every path
    struct restore { int d; ~restore() { (void) hipSetDevice(d); } } rs_{ dev_before };
    HIP_CHECK(hipSetDevice(cs->device)); mi::flush_uploads(cs);
    if (cd != cs) { HIP_CHECK(hipSetDevice(cd->device)); mi::flush_uploads(cd); }
    // RCCL between different devices (the real case), or inside ONE backend (self send / recv on one stream: how a 1-GPU box exercises the
    // RCCL plumbing).  Two backends on the same device would put the two halves of a self-exchange on different streams of one communicator,
    // which NCCL's group semantics do not allow -- that case is a plain device copy.
    const bool rccl = (cs->device != cd->device || cs == cd) && mi::handoff_init_locked() == 1 && cs->device < (int) mi::g_comms.size() && cd->device < (int) mi::g_comms.size();
    if (rccl) {
        int rc = mi::g_rccl.GroupStart();
        HIP_CHECK(hipSetDevice(cs->device));
        if (!rc) rc = mi::g_rccl.Send(src, nbytes, /*ncclInt8*/ 0, cd->device, mi::g_comms[cs->device], cs->stream);
        HIP_CHECK(hipSetDevice(cd->device));
        if (!rc) rc = mi::g_rccl.Recv(dst, nbytes, /*ncclInt8*/ 0, cs->device, mi::g_comms[cd->device], cd->stream);
        const int rc2 = mi::g_rccl.GroupEnd();
        if (rc || rc2) { mi::log_msg(GGML_LOG_LEVEL_ERROR, "[mi355x] hand-off: RCCL send/recv failed (%s)\n", mi::g_rccl.GetErrorString(rc ? rc : rc2)); return -2; }
        ++mi::g_ho_rccl_calls;
        return 1;
    }
    // fallback: peer copy on the source stream, the destination stream waits for it
    HIP_CHECK(hipSetDevice(cs->device));
    if (cs->device == cd->device) HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, cs->stream));
    else                          HIP_CHECK(hipMemcpyPeerAsync(dst, cd->device, src, cs->device, nbytes, cs->stream));
    if (cs != cd) {
        if (!cs->handoff_event) HIP_CHECK(hipEventCreateWithFlags(&cs->handoff_event, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(cs->handoff_event, cs->stream));
        HIP_CHECK(hipSetDevice(cd->device));
        HIP_CHECK(hipStreamWaitEvent(cd->stream, cs->handoff_event, 0));
    }
    ++mi::g_ho_peer_calls;
    return 2;
}

int mi355x_handoff_tensor(struct ggml_backend * src_backend, const struct ggml_tensor * src, struct ggml_backend * dst_backend, struct ggml_tensor * dst) {
    if (!src || !dst || !src->data || !dst->data) return -1;
    size_t ns = 1, nd = 1;
    for (int i = 0; i < 4; ++i) { ns *= (size_t) src->ne[i]; nd *= (size_t) dst->ne[i]; }
    // dense f32 / f16 rows only (what the hidden-state chunk is): same element count and type, both contiguous
    if (src->type != dst->type || ns != nd || (src->type != GGML_TYPE_F32 && src->type != GGML_TYPE_F16)) return -1;
    const size_t es = src->type == GGML_TYPE_F32 ? 4 : 2;
    size_t st = es, dt = es;
    for (int i = 0; i < 4; ++i) { if (src->ne[i] > 1 && src->nb[i] != st) return -1; if (dst->ne[i] > 1 && dst->nb[i] != dt) return -1; st *= (size_t) src->ne[i]; dt *= (size_t) dst->ne[i]; }
    return mi355x_handoff(src_backend, src->data, dst_backend, dst->data, ns * es);
}

long mi355x_handoff_count(int kind) { std::lock_guard<std::mutex> lk(mi::g_ho_mu); return kind == 1 ? mi::g_ho_rccl_calls : mi::g_ho_peer_calls; }

void mi355x_handoff_shutdown(void) {
    std::lock_guard<std::mutex> lk(mi::g_ho_mu);
    if (mi::g_ho_state == 1) for (mi::nccl_comm_t c : mi::g_comms) if (c) (void) mi::g_rccl.CommDestroy(c);
    mi::g_comms.clear(); mi::g_ho_state = 0;
}

// Module -> device pinning of the omni pipeline (BASELINE configs[3], [4]): which MI355X of the node a GGUF module lives on.  Default map
// (one module per GPU, in pipeline order): vpm 0, apm 1, llm 2, tts 3, t2w (Token2Wav flow / DiT) 4, vocoder 5 -- taken modulo the number of
// visible devices, so a 3-GPU run pins VPM / APM / LLM to three GPUs (C4) and a 1-GPU run puts everything on device 0.  Override with
// MI355X_MODULE_MAP="llm=0,tts=1,...".  Returns the device ordinal (the index of "MI355X<i>" in the registry), or -1 for an unknown name.
int mi355x_module_device(const char * module) {
    if (!module) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void) hipGetLastError(); n = 1; }
    static const char * names[] = { "vpm", "apm", "llm", "tts", "t2w", "vocoder" };
    int idx = -1;
    for (int i = 0; i < 6; ++i) if (!strcmp(module, names[i])) idx = i;
    if (idx < 0) return -1;
    if (const char * map = getenv("MI355X_MODULE_MAP")) {
        const std::string m(map), key = std::string(module) + "=";
        size_t p = 0;
        while (p < m.size()) {
            size_t e = m.find(',', p); if (e == std::string::npos) e = m.size();
            if (m.compare(p, key.size(), key) == 0) { const int d = atoi(m.c_str() + p + key.size()); return d >= 0 && d < n ? d : -1; }
            p = e + 1;
        }
    }
    return idx % n;
}

} // extern "C"
