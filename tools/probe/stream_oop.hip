// Does the Adam stream (3 reads + 3 writes per element, fp32) run faster OUT of place (read p/m/v, write p2/m2/v2:
// ping-pong slabs) than in place? Plus the pure-read and pure-write ceilings of the same geometry
// (strips of 128 columns, U = 8 rows in flight: the geometry of linear_wgrad_adam_dgrad_kernel).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: in place, 1: out of place, 2: read only (3 arrays, result reduced into one float4 per thread), 3: write only
template <int U, int MODE>
__global__ __launch_bounds__(256) void k(float* p, float* m, float* v, float* p2, float* m2, float* v2, long cs, int M, int N,
                                         int rows_per_block) {
    constexpr int TPR = 32, R = 8;
    const int c = blockIdx.z;
    const int col = blockIdx.x * 128 + (threadIdx.x % TPR) * 4;
    const int row0 = blockIdx.y * rows_per_block + threadIdx.x / TPR;
    const size_t base = (size_t)c * cs + (size_t)row0 * N + col;
    const int steps = rows_per_block / R;
    f32x4 acc = {0, 0, 0, 0};
    for (int s0 = 0; s0 < steps; s0 += U) {
        f32x4 pv[U], mv[U], vv[U];
        if (MODE != 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = base + (size_t)(s0 + u) * R * N;
                pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + o));
                mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + o));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + o));
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) { pv[u] = acc; mv[u] = acc; vv[u] = acc; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = base + (size_t)(s0 + u) * R * N;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = pv[u][j] * 1e-3f;
                mv[u][j] = 0.9f * mv[u][j] + 0.1f * g;
                vv[u][j] = 0.999f * vv[u][j] + 0.001f * g * g;
                pv[u][j] -= 1e-3f * (mv[u][j] / (sqrtf(vv[u][j]) + 1e-8f));
            }
            if (MODE == 2) { acc += pv[u] + mv[u] + vv[u]; continue; }
            float* dp = MODE == 1 ? p2 : p; float* dm = MODE == 1 ? m2 : m; float* dv = MODE == 1 ? v2 : v;
            __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(dp + o));
            __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(dm + o));
            __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(dv + o));
        }
    }
    if (MODE == 2 && acc[0] == 123.456f) *reinterpret_cast<f32x4*>(p2 + base) = acc;     // keeps the loads alive
}

template <int U, int MODE>
static void run(const char* name, float** a, long cs, int K, int M, int N, int rpb, bool swap) {
    dim3 grid(N / 128, M / rpb, K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int i) {
        const int s = (swap && (i & 1)) ? 3 : 0;       // ping-pong: every other launch reads what the last one wrote
        hipLaunchKernelGGL((k<U, MODE>), grid, dim3(256), 0, 0, a[s], a[s + 1], a[s + 2], a[3 - s], a[4 - s], a[5 - s], cs, M, N, rpb);
    };
    for (int i = 0; i < 2; ++i) launch(i);
    hipEventRecord(e0, 0);
    const int reps = 6;
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double words = MODE == 2 || MODE == 3 ? 3.0 : 6.0;
    const double gb = words * K * (double)M * N * 4 / 1e9;
    printf("%-44s U %2d rows/block %4d: %.3f ms  %.2f TB/s\n", name, U, rpb, ms, gb / ms);
    fflush(stdout);
}

int main() {
    const int K = 100, M = 2048, N = 3072;
    const long cs = (long)M * N;
    float* a[6];
    for (int i = 0; i < 6; ++i) { if (hipMalloc(&a[i], K * cs * 4) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(a[i], 0, K * cs * 4); }
    for (int rpb : {2048, 128}) {
        run<8, 0>("in place (3r + 3w)", a, cs, K, M, N, rpb, false);
        run<8, 1>("out of place, ping-pong (3r + 3w)", a, cs, K, M, N, rpb, true);
        run<8, 1>("out of place, same direction (3r + 3w)", a, cs, K, M, N, rpb, false);
        run<4, 1>("out of place, ping-pong (3r + 3w)", a, cs, K, M, N, rpb, true);
        run<8, 2>("read only (3r)", a, cs, K, M, N, rpb, false);
        run<8, 3>("write only (3w)", a, cs, K, M, N, rpb, false);
    }
    return 0;
}
