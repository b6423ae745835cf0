// kernels/act_dev.hpp -- activation functions shared by the element-wise kernels and the split-K reduction's epilogue (device code only)
#pragma once
#include "../kernels.hpp"

namespace mi {

// GELU / GELU_QUICK of an f32 value go through the reference's f16 tables (GGML_GELU_FP16 / GGML_GELU_QUICK_FP16, vec.h:17-18, :892-906, :933-941;
// tables filled in ggml-cpu.c:3555-3556 with f16(ggml_gelu_f32(f)) for every f16 value f): the argument is rounded to f16, the formula evaluated in
// f32 and the result rounded to f16 -- evaluated here instead of looked up; GELU short-cuts x <= -10 to 0 and x >= 10 to x before the table.
static __device__ __forceinline__ float op_gelu(float x) {
    if (x <= -10.0f) return 0.0f;
    if (x >= 10.0f)  return x;
    const float f = h2f(f2h(x));
    return h2f(f2h(0.5f * f * (1.0f + tanhf(0.79788456080286535587989211986876f * f * (1.0f + 0.044715f * f * f)))));
}
static __device__ __forceinline__ float op_gelu_quick(float x) {
    const float f = h2f(f2h(x));
    return h2f(f2h(f * (1.0f / (1.0f + expf(-1.702f * f)))));
}

} // namespace mi
