// fattn_dev.hpp -- device-side argument block shared by the two FLASH_ATTN_EXT kernels (fattn.hip: decode, fattn_mma.hip: prefill)
#pragma once
#include "../kernels.hpp"
#include "norm_rope_dev.hpp"

namespace mi {

// decode pre-stage (qraw == null: none): the layer's q / k chains (RMS_NORM -> MUL -> ROPE) and the k / v cache stores are done by
// the attention workgroup of each KV head itself, so the separate k_norm_rope launch (and its ~5 us of dependency latency)
// disappears; arithmetic is norm_rope_wave's, shared with that kernel
struct fa_pre {
    const char * qraw; int64_t q_hs; const char * kraw; int64_t k_hs; const char * vraw; int64_t v_hs;   // f32 [D] per head, head strides in bytes
    const float * qw; const float * kw; const int32_t * pos; const float * ff; float eps; rope_dev rd;
    char * kcache; int64_t kc_rs; char * vcache; int64_t vc_rs; const char * kidx; const char * vidx; int idx_is64;
};

struct fa_dev {
    const char * q; const char * k; const char * v; const char * mask; const float * sinks; char * dst;
    int     nq, nh, nhkv, nkv, ns;                      // query rows, heads, kv heads, kv length, sequences
    int64_t qnb1, qnb2, qnb3, knb1, knb2, knb3, vnb1, vnb2, vnb3;
    int64_t mnb1, mnb2, mnb3, mne2, mne3;
    int64_t dnb1, dnb2, dnb3;
    float scale, max_bias, logit_softcap, m0, m1; uint32_t n_head_log2;
    int gq;                                             // n_head / n_head_kv
    int hpw;                                            // heads handled per workgroup (<= R)
    int qpw;                                            // query rows per workgroup (R / hpw)
    const uint8_t * tile_map; int map_nqb;              // mask tile classes [mne3][mne2][map_nqb][ntile] (prefill kernel), or null = no mask
    int nsplit; float * part;                           // decode kernel: KV range cut into nsplit workgroups per head group, partial (O, M, S) rows in `part`
    unsigned * cnt = nullptr;                           // one-token kernel with slices: per-head arrival counters (merge inside the launch)
    fa_pre pre;
    char * out16; int64_t out16_rs; int write_f32;      // prefill kernel: f16 copy of the output rows (activation image of wo's GEMM)
    char * img; size_t img_bytes;                       // optional Q8_K image output (one image per (seq, query row)), else null
    unsigned long long * stamps = nullptr;              // measurement only (MI355X_FA_STAMPS): s_memrealtime marks of workgroup 0, wave 0
    int vt = 0;                                         // prefill kernel: v is V^T ([n_kv] cells contiguous per d row, vnb1 = row stride); value = the alignment every row start has (16 / 4 / 2)
};

// prefill kernel (fattn_mma.hip): MFMA tiles, 32 query rows per wave
void flash_attn_ext_mma(const fa_dev & a, int D, hipStream_t st);
void fattn_mask_map(const fa_dev & a, uint8_t * map, hipStream_t st);     // classify the mask's 32 x 32 tiles
bool fattn_mma_ok(int64_t nkv);
// decode shape on the matrix cores (k_fattn_gqa): a.qpw tokens x gq heads of a KV head (<= 32 pairs) as one query tile, nw = 4 or 8 waves,
// KV range in a.nsplit slices (-> a.part when > 1), optional a.pre / a.img
void flash_attn_ext_gqa(const fa_dev & a, int D, int nw, hipStream_t st);
// one token of one sequence over <= 256 cache rows with the pre-stage (fattn_one.hip); rope_tab: the token's (cos, sin) table
void flash_attn_one(const fa_dev & a, int D, const float * rope_tab, hipStream_t st);
// the same step as one workgroup per (KV head, 64-row slice) leaving partial (O, M, S) states in `parts` (fattn_gs_parts_bytes) for the wo launch's prologue to fold
void flash_attn_gs(const fa_dev & a, int D, const float * rope_tab, float * parts, hipStream_t st);
// any other head size (fattn_any.hip): one wave per (query row, head, sequence)
bool fattn_any_ok(int64_t Dk, int64_t Dv);
void flash_attn_ext_any(const fa_dev & a, int Dk, int Dv, int kv_type, hipStream_t st);      // kv_type: F16 / F32 / BF16 / Q8_0 / Q4_0 (K and V alike)

} // namespace mi
