// Which access geometry lets an in-place 3-read / 3-write fp32 stream (Adam: p, m, v) reach the HBM ceiling?
// Blocks walk down the rows of a [M][N] matrix per client in strips of COLS columns; U row-steps of loads are
// in flight per thread. COLS = 128 is the geometry of linear_wgrad_adam_dgrad_kernel (8 rows x 512 B per step).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int COLS, int U, bool NT>
__global__ __launch_bounds__(256) void strip_update(float* p, float* m, float* v, long cs, int M, int N, int rows_per_block) {
    constexpr int TPR = COLS / 4;          // threads per row
    constexpr int R = 256 / TPR;           // rows per step
    const int c = blockIdx.z;
    const int col = blockIdx.x * COLS + (threadIdx.x % TPR) * 4;
    const int row0 = blockIdx.y * rows_per_block + threadIdx.x / TPR;
    const size_t base = (size_t)c * cs + (size_t)row0 * N + col;
    const int steps = rows_per_block / R;
    auto ld = [&](const float* q, size_t o) {
        return NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + o)) : *reinterpret_cast<const f32x4*>(q + o);
    };
    auto st = [&](float* q, size_t o, f32x4 x) {
        if (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(q + o)); else *reinterpret_cast<f32x4*>(q + o) = x;
    };
    for (int s0 = 0; s0 < steps; s0 += U) {
        f32x4 pv[U], mv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = base + (size_t)(s0 + u) * R * N;
            pv[u] = ld(p, o); mv[u] = ld(m, o); vv[u] = ld(v, o);
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
            st(p, o, pv[u]); st(m, o, mv[u]); st(v, o, vv[u]);
        }
    }
}

template <int COLS, int U, bool NT>
static void run(const char* name, float* p, float* m, float* v, long cs, int K, int M, int N, int rpb) {
    dim3 grid(N / COLS, M / rpb, K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((strip_update<COLS, U, NT>), grid, dim3(256), 0, 0, p, m, v, cs, M, N, rpb);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((strip_update<COLS, U, NT>), grid, dim3(256), 0, 0, p, m, v, cs, M, N, rpb);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double gb = 6.0 * K * (double)M * N * 4 / 1e9;
    printf("%-34s cols %4d U %d nt %d rows/block %4d blocks %6d: %.3f ms  %.2f TB/s\n", name, COLS, U, (int)NT, rpb,
           grid.x * grid.y * grid.z, ms, gb / ms);
    fflush(stdout);
}

// MODE 1: m and v interleaved per float4 group ([m4 v4] = 32 B per thread); MODE 2: p, m, v interleaved (48 B per thread)
template <int COLS, int U, int MODE>
__global__ __launch_bounds__(256) void strip_update_il(float* p, float* mv, long cs, int M, int N, int rows_per_block) {
    constexpr int TPR = COLS / 4;
    constexpr int R = 256 / TPR;
    const int c = blockIdx.z;
    const int col = blockIdx.x * COLS + (threadIdx.x % TPR) * 4;
    const int row0 = blockIdx.y * rows_per_block + threadIdx.x / TPR;
    const size_t base = (size_t)c * cs + (size_t)row0 * N + col;
    const int steps = rows_per_block / R;
    for (int s0 = 0; s0 < steps; s0 += U) {
        f32x4 pv[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = base + (size_t)(s0 + u) * R * N;
            if (MODE == 1) {
                pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + o));
                mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mv + 2 * o));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(mv + 2 * o + 4));
            } else {
                pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 3 * o));
                mm[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 3 * o + 4));
                vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 3 * o + 8));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t o = base + (size_t)(s0 + u) * R * N;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = pv[u][j] * 1e-3f;
                mm[u][j] = 0.9f * mm[u][j] + 0.1f * g;
                vv[u][j] = 0.999f * vv[u][j] + 0.001f * g * g;
                pv[u][j] -= 1e-3f * (mm[u][j] / (sqrtf(vv[u][j]) + 1e-8f));
            }
            if (MODE == 1) {
                __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + o));
                __builtin_nontemporal_store(mm[u], reinterpret_cast<f32x4*>(mv + 2 * o));
                __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(mv + 2 * o + 4));
            } else {
                __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + 3 * o));
                __builtin_nontemporal_store(mm[u], reinterpret_cast<f32x4*>(p + 3 * o + 4));
                __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(p + 3 * o + 8));
            }
        }
    }
}

template <int COLS, int U, int MODE>
static void run_il(const char* name, float* p, float* mv, long cs, int K, int M, int N, int rpb) {
    dim3 grid(N / COLS, M / rpb, K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((strip_update_il<COLS, U, MODE>), grid, dim3(256), 0, 0, p, mv, cs, M, N, rpb);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((strip_update_il<COLS, U, MODE>), grid, dim3(256), 0, 0, p, mv, cs, M, N, rpb);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double gb = 6.0 * K * (double)M * N * 4 / 1e9;
    printf("%-34s cols %4d U %d mode %d rows/block %4d blocks %6d: %.3f ms  %.2f TB/s\n", name, COLS, U, MODE, rpb,
           grid.x * grid.y * grid.z, ms, gb / ms);
    fflush(stdout);
}

// loads and stores grouped BY ARRAY (U rows of p, then U rows of m, then U rows of v) instead of interleaved per row
template <int COLS, int U>
__global__ __launch_bounds__(256) void strip_update_grouped(float* p, float* m, float* v, long cs, int M, int N, int rows_per_block) {
    constexpr int TPR = COLS / 4, R = 256 / TPR;
    const int c = blockIdx.z;
    const int col = blockIdx.x * COLS + (threadIdx.x % TPR) * 4;
    const int row0 = blockIdx.y * rows_per_block + threadIdx.x / TPR;
    const size_t base = (size_t)c * cs + (size_t)row0 * N + col;
    const int steps = rows_per_block / R;
    for (int s0 = 0; s0 < steps; s0 += U) {
        f32x4 pv[U], mv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + base + (size_t)(s0 + u) * R * N));
#pragma unroll
        for (int u = 0; u < U; ++u) mv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m + base + (size_t)(s0 + u) * R * N));
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v + base + (size_t)(s0 + u) * R * N));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float g = pv[u][j] * 1e-3f;
                mv[u][j] = 0.9f * mv[u][j] + 0.1f * g;
                vv[u][j] = 0.999f * vv[u][j] + 0.001f * g * g;
                pv[u][j] -= 1e-3f * (mv[u][j] / (sqrtf(vv[u][j]) + 1e-8f));
            }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(pv[u], reinterpret_cast<f32x4*>(p + base + (size_t)(s0 + u) * R * N));
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m + base + (size_t)(s0 + u) * R * N));
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v + base + (size_t)(s0 + u) * R * N));
    }
}
template <int COLS, int U>
static void run_grouped(const char* name, float* p, float* m, float* v, long cs, int K, int M, int N, int rpb) {
    dim3 grid(N / COLS, M / rpb, K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((strip_update_grouped<COLS, U>), grid, dim3(256), 0, 0, p, m, v, cs, M, N, rpb);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((strip_update_grouped<COLS, U>), grid, dim3(256), 0, 0, p, m, v, cs, M, N, rpb);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double gb = 6.0 * K * (double)M * N * 4 / 1e9;
    printf("%-34s cols %4d U %d rows/block %4d blocks %6d: %.3f ms  %.2f TB/s\n", name, COLS, U, rpb, grid.x * grid.y * grid.z, ms, gb / ms);
    fflush(stdout);
}

int main() {
    const int K = 100, M = 2048, N = 3072;
    const long cs = (long)M * N;
    float *p, *m, *v;
    hipMalloc(&p, K * cs * 4); hipMalloc(&m, K * cs * 4); hipMalloc(&v, K * cs * 4);
    hipMemset(p, 0, K * cs * 4); hipMemset(m, 0, K * cs * 4); hipMemset(v, 0, K * cs * 4);
    run<128, 1, true>("strip 512 B, serial", p, m, v, cs, K, M, N, 2048);
    run<128, 2, true>("strip 512 B", p, m, v, cs, K, M, N, 2048);
    run<128, 4, true>("strip 512 B", p, m, v, cs, K, M, N, 2048);
    run<128, 8, true>("strip 512 B", p, m, v, cs, K, M, N, 2048);
    run<128, 4, false>("strip 512 B", p, m, v, cs, K, M, N, 2048);
    run<128, 4, true>("strip 512 B, 128-row blocks", p, m, v, cs, K, M, N, 128);
    run<256, 4, true>("strip 1 KB", p, m, v, cs, K, M, N, 2048);
    run<256, 4, true>("strip 1 KB, 256-row blocks", p, m, v, cs, K, M, N, 256);
    run<512, 4, true>("strip 2 KB", p, m, v, cs, K, M, N, 2048);
    run<512, 4, true>("strip 2 KB, 512-row blocks", p, m, v, cs, K, M, N, 512);
    run<512, 4, false>("strip 2 KB, 512-row blocks", p, m, v, cs, K, M, N, 512);
    run<1024, 4, true>("strip 4 KB, 256-row blocks", p, m, v, cs, K, M, N, 256);
    run<1024, 4, false>("strip 4 KB, 256-row blocks", p, m, v, cs, K, M, N, 256);
    run<1024, 2, true>("strip 4 KB, 64-row blocks", p, m, v, cs, K, M, N, 64);
    run<1024, 4, true>("strip 4 KB, 16-row blocks", p, m, v, cs, K, M, N, 16);
    run<1024, 4, false>("strip 4 KB, 16-row blocks", p, m, v, cs, K, M, N, 16);
    run_grouped<128, 8>("grouped by array, 512 B", p, m, v, cs, K, M, N, 2048);
    run_grouped<128, 8>("grouped by array, 512 B 128-row", p, m, v, cs, K, M, N, 128);
    run_grouped<1024, 8>("grouped by array, 4 KB 64-row", p, m, v, cs, K, M, N, 64);
    run_grouped<1024, 16>("grouped by array, 4 KB 64-row", p, m, v, cs, K, M, N, 64);
    run_grouped<128, 16>("grouped by array, 512 B", p, m, v, cs, K, M, N, 2048);
    // interleaved layouts: p is the first third / p + cs*K.. (the three allocations are used as one pool here)
    float* pool; hipMalloc(&pool, 3 * K * cs * 4); hipMemset(pool, 0, 3 * K * cs * 4);
    run_il<128, 4, 1>("w + [m v] interleaved, 512 B", pool, pool + K * cs, cs, K, M, N, 2048);
    run_il<128, 4, 1>("w + [m v] interleaved, 128-row", pool, pool + K * cs, cs, K, M, N, 128);
    run_il<1024, 4, 1>("w + [m v] interleaved, 4 KB 16-row", pool, pool + K * cs, cs, K, M, N, 16);
    run_il<128, 4, 2>("[w m v] interleaved, 512 B", pool, pool, cs, K, M, N, 2048);
    run_il<128, 4, 2>("[w m v] interleaved, 128-row", pool, pool, cs, K, M, N, 128);
    run_il<1024, 4, 2>("[w m v] interleaved, 4 KB 16-row", pool, pool, cs, K, M, N, 16);
    // do the three streams collide in the DRAM banks? shift m and v against p by various byte offsets
    {
        float* big; hipMalloc(&big, (3 * K * cs + (64 << 20)) * 4); hipMemset(big, 0, (3 * K * cs + (64 << 20)) * 4);
        const long deltas[] = {0, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096 + 64, 3 << 20};
        for (long d : deltas) {
            char name[64]; snprintf(name, sizeof name, "m,v shifted by %ld B / 2x", d * 4);
            run<128, 4, true>(name, big, big + K * cs + d, big + 2 * K * cs + 2 * d, cs, K, M, N, 128);
        }
        // one array only (1r 1w in place), 3x the rows for the same bytes: is it the three streams or the kernel?
        printf("one array, 3x bytes per launch normalised: ");
        run<128, 4, true>("p only x3 (same pointer)", big, big, big, cs, K, M, N, 128);
    }
    return 0;
}
