// Synthetic co-runners for tools/overlap_probe.py: register-only loops, one wave per SIMD, no memory, no LDS.
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

extern "C" __global__ __launch_bounds__(256) void burn_mfma_f32(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = seed + threadIdx.x, b = seed * 0.5f;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[0] = s;
}
extern "C" __global__ __launch_bounds__(256) void burn_mfma_bf16(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed * 0.5f); }
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 12345.678f) out[0] = s;
}
extern "C" __global__ __launch_bounds__(256) void burn_valu(float* out, int iters, float seed) {
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = seed + e + threadIdx.x;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = __builtin_fmaf(v[e], 1.0001f, 0.5f);
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += v[e];
    if (s == 12345.678f) out[0] = s;
}
extern "C" __global__ __launch_bounds__(256) void burn_sleep(float* out, int iters, float seed) {
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    if (seed == 12345.678f) out[0] = seed;
}
// occupies a whole CU (dynamic LDS of ~150 KB: nothing else fits beside it) and does nothing: how does the HBM-bound
// update kernel fare on the CUs that are left?
extern "C" __global__ __launch_bounds__(64) void burn_hog(float* out, int iters, float seed) {
    extern __shared__ float lds[];
    if (seed == 12345.678f) lds[threadIdx.x] = seed;
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(64);
    if (seed == 12345.678f) out[0] = lds[threadIdx.x];
}
extern "C" int burn_hog_launch(int blocks, int iters, float* out, hipStream_t s) {
    hipFuncSetAttribute((const void*)burn_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(burn_hog, dim3(blocks), dim3(64), 150 * 1024, s, out, iters, 1.0f);
    return (int)hipGetLastError();
}
extern "C" int burn_launch(int which, int blocks, int iters, float* out, hipStream_t s) {
    if (which == 0) hipLaunchKernelGGL(burn_mfma_f32, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f);
    else if (which == 1) hipLaunchKernelGGL(burn_mfma_bf16, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f);
    else if (which == 2) hipLaunchKernelGGL(burn_valu, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f);
    else hipLaunchKernelGGL(burn_sleep, dim3(blocks), dim3(256), 0, s, out, iters, 1.0f);
    return (int)hipGetLastError();
}
// where does a block run? out[2*b] = XCC id, out[2*b+1] = HW_ID (cu_id bits 11:8, sh 12, se 15:13)
extern "C" __global__ void whereami(unsigned* out, int spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
extern "C" int whereami_launch(int blocks, int spin, unsigned* out, hipStream_t s) {
    hipLaunchKernelGGL(whereami, dim3(blocks), dim3(64), 0, s, out, spin);
    return (int)hipGetLastError();
}
// stream restricted to the CUs of a bit mask (hipExtStreamCreateWithCUMask)
extern "C" int masked_stream_create(hipStream_t* s, int words, const unsigned* mask) {
    return (int)hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask);
}
