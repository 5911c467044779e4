// Which lane -> address patterns of ds_read_b128 are bank-conflict-free on gfx950? One wave per block reads 16 bytes
// per lane at  base(lane) = (lane & 31) * stride + (lane >> 5) * hoff  many times; cycles per read are printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(64) void probe(int stride, int hoff, int mode, int iters, float* out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 16; i += 64) reinterpret_cast<f32x4*>(lds)[i] = f32x4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    const int lane = threadIdx.x;
    int addr;
    if (mode == 0) addr = (lane & 31) * stride + (lane >> 5) * hoff;           // 32 rows, two k halves
    else if (mode == 1) addr = lane * stride;                                  // 64 rows
    else { const int m = lane & 31; addr = (m % 14) * stride + (m / 14) * 64 + (lane >> 5) * hoff; }   // column-major pixels
    addr &= 65535 - 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4 v;
            const unsigned a32 = (unsigned)(size_t)(lds) + ((addr + u * 4096) & 65535);
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a32));
            acc += v;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (acc[0] == 12345.f) out[0] = acc[1];
    if (lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 64); hipMallocManaged(&cyc, 8);
    struct { const char* name; int stride, hoff, mode; } cases[] = {
        {"linear 16 B (lane * 16)", 16, 0, 1},
        {"rows 80 B apart + 16 B half (GEMM tiles)", 80, 16, 0},
        {"rows 64 B apart + 16 B half (row-major pixels)", 64, 16, 0},
        {"rows 144 B apart + 16 B half", 144, 16, 0},
        {"rows 1168 B apart + 16 B half (image rows 5x5)", 1168, 16, 0},
        {"rows 1040 B apart + 16 B half (image rows 3x3)", 1040, 16, 0},
        {"column-major pixels, row 1168 B", 1168, 16, 2},
        {"column-major pixels, row 1040 B", 1040, 16, 2},
        {"rows 256 B apart (worst case)", 256, 16, 0},
        {"rows 128 B apart", 128, 16, 0},
        {"rows 32 B apart + 16 B half", 32, 16, 0},
        {"rows 64 B apart + 32 B half", 64, 32, 0},
        {"rows 80 B apart, 64 rows", 80, 0, 1},
    };
    const int iters = 2000;
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.stride, c.hoff, c.mode, iters, out, cyc);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c.stride, c.hoff, c.mode, iters, out, cyc);
        hipDeviceSynchronize();
        printf("%-52s %6.1f cycles per ds_read_b128\n", c.name, (double)cyc[0] / (iters * 8.0));
    }
    return 0;
}
