// FedAvg as one fused weighted row-reduce over the parameter slab.
//
//   for every aggregation group g:   out[d] = sum_s wn[s] * row[src[s]]      for all d in dst(g)
//
// A group is one distinct (contributor set, weights) pair: in a fully connected federation
// every node computes the same average, so there is ONE group with n destinations; in a
// ring/random topology there is one group per node; partial aggregations of the gossip
// protocol (reference: fedstellar/learning/aggregators/aggregator.py:181-208) are groups over
// subsets. Weights are pre-normalised on the host (w_i / sum w), accumulation is fp32, every
// source row is read exactly once per group: the kernel is a pure HBM stream (float4 / lane).
//
// Source rows live either in the local slab (row id < n_local) or in the staging slab that
// receives rows of other GPUs over RCCL (row id >= n_local).
//
// Replaces the Python double loop of fedstellar/learning/aggregators/fedavg.py:40-60.
#include <cstdlib>

#include "common.h"

namespace {

struct FedAvgArgs {
    const float* local;  long local_cs;  int n_local;
    const float* remote; long remote_cs;
    float* out;          long out_cs;
    const int* src_ptr;   // [G+1]
    const int* src_row;   // [nnz]
    const float* src_w;   // [nnz] normalised weights
    const int* dst_ptr;   // [G+1]
    const int* dst_row;   // [ndst] rows of `out`
    long n4;              // float4 per row
};

__device__ __forceinline__ const float4* row_ptr(const FedAvgArgs& a, int r) {
    return reinterpret_cast<const float4*>(r < a.n_local ? a.local + (long)r * a.local_cs
                                                         : a.remote + (long)(r - a.n_local) * a.remote_cs);
}

__global__ __launch_bounds__(256) void fedavg_kernel(FedAvgArgs a) {
    const int g = blockIdx.y;
    const int s0 = a.src_ptr[g], s1 = a.src_ptr[g + 1];
    const int d0 = a.dst_ptr[g], d1 = a.dst_ptr[g + 1];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (long)gridDim.x * 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = s0;
        for (; s + 4 <= s1; s += 4) {   // 4 independent 16-B loads in flight per lane
            const float4 v0 = row_ptr(a, a.src_row[s])[i];
            const float4 v1 = row_ptr(a, a.src_row[s + 1])[i];
            const float4 v2 = row_ptr(a, a.src_row[s + 2])[i];
            const float4 v3 = row_ptr(a, a.src_row[s + 3])[i];
            const float w0 = a.src_w[s], w1 = a.src_w[s + 1], w2 = a.src_w[s + 2], w3 = a.src_w[s + 3];
            acc.x += w0 * v0.x + w1 * v1.x + w2 * v2.x + w3 * v3.x;
            acc.y += w0 * v0.y + w1 * v1.y + w2 * v2.y + w3 * v3.y;
            acc.z += w0 * v0.z + w1 * v1.z + w2 * v2.z + w3 * v3.z;
            acc.w += w0 * v0.w + w1 * v1.w + w2 * v2.w + w3 * v3.w;
        }
        for (; s < s1; ++s) {
            const float4 v0 = row_ptr(a, a.src_row[s])[i];
            const float w0 = a.src_w[s];
            acc.x += w0 * v0.x; acc.y += w0 * v0.y; acc.z += w0 * v0.z; acc.w += w0 * v0.w;
        }
        for (int d = d0; d < d1; ++d)
            reinterpret_cast<float4*>(a.out + (long)a.dst_row[d] * a.out_cs)[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// In-place variant: out == local. A block owns a column chunk of C floats for ALL rows: it first pulls the chunk of
// every source row (local + staged) into LDS - each row is read from HBM exactly ONCE however many groups use it
// (a random 100-node topology reads every row 5.1 times in the kernel above: 13.4 GB instead of 2.6 GB) - and after
// a barrier writes every destination row's chunk straight back into the slab. Column chunks are disjoint between
// blocks, so nothing a block overwrites is still needed by another: no second slab, no copy back (the captured
// training graphs hold pointers into the one parameter slab). Rows no group writes simply keep their model.
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void fedavg_inplace_kernel(FedAvgArgs a, int n_rows, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) float pool[];      // [n_rows][C]
    constexpr int C4 = C / 4, RPP = 256 / C4;                          // float4 per row chunk, rows per pass
    const long col4 = (long)blockIdx.x * C4;
    const int lc = threadIdx.x % C4, lr = threadIdx.x / C4;
    const bool col_ok = col4 + lc < a.n4;
    constexpr int U = 8;                     // row pieces in flight per thread
    for (int r0 = 0; r0 < n_rows; r0 += RPP * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * RPP + lr;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < n_rows && col_ok) {
                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(row_ptr(a, r) + col4 + lc));
                v[u] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r0 + u * RPP + lr;
            if (r < n_rows) reinterpret_cast<float4*>(pool)[r * C4 + lc] = v[u];
        }
    }
    __syncthreads();
    for (int g = lr; g < n_groups; g += RPP) {
        const int s0 = a.src_ptr[g], s1 = a.src_ptr[g + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = s0; s < s1; ++s) {
            const float4 x = reinterpret_cast<const float4*>(pool)[a.src_row[s] * C4 + lc];
            const float w = a.src_w[s];
            acc.x += w * x.x; acc.y += w * x.y; acc.z += w * x.z; acc.w += w * x.w;
        }
        if (col_ok)
            for (int d = a.dst_ptr[g]; d < a.dst_ptr[g + 1]; ++d)
                reinterpret_cast<float4*>(a.out + (long)a.dst_row[d] * a.out_cs)[col4 + lc] = acc;
    }
}

}  // namespace

// LDS bytes the in-place kernel may use per block (two to three blocks per CU)
// LDS bytes per block the in-place kernel aims at: 32 KB = four to five blocks per CU, whose load and store phases
// cover each other (100 rows x 6.5 M on MI355X: 16 KB 1.37 ms, 32 KB 1.20 ms, 64 KB 1.57 ms; tools/probe/fedavg_probe.py)
static long fedavg_inplace_lds() {
    static const long v = [] { const char* e = getenv("P2PFL_FEDAVG_LDS_KB"); return (e ? atol(e) : 32) * 1024; }();
    return v;
}

// Column chunk (floats) of the in-place reduce for n_rows source rows, 0: not applicable (too many rows for 128-byte
// row pieces in 64 KB of LDS: use p2pfl_fedavg into a second slab)
P2PFL_API int p2pfl_fedavg_inplace_chunk(int n_rows) {
    for (int c = 256; c >= 32; c >>= 1)
        if ((long)n_rows * c * 4 <= fedavg_inplace_lds()) return c;
    return (long)n_rows * 32 * 4 <= 64 * 1024 ? 32 : 0;
}

// slab[dst] = sum w * pool[src], pool = slab rows [0, n_local) then `remote` rows [0, n_remote); see the kernel
P2PFL_API int p2pfl_fedavg_inplace(float* slab, long slab_cs, int n_local, const float* remote, long remote_cs,
                                   int n_remote, const int* src_ptr, const int* src_row, const float* src_w,
                                   const int* dst_ptr, const int* dst_row, int n_groups, long n, hipStream_t stream) {
    if (n_groups <= 0 || n <= 0 || n_local <= 0 || n_remote < 0) return -1;
    if ((n & 3) || (slab_cs & 3) || (remote_cs & 3) || ((uintptr_t)slab & 15) || ((uintptr_t)remote & 15)) return -2;
    const int rows = n_local + n_remote;
    const int C = p2pfl_fedavg_inplace_chunk(rows);
    if (C == 0) return -4;
    FedAvgArgs a{slab, slab_cs, n_local, remote ? remote : slab, remote_cs, slab, slab_cs, src_ptr, src_row, src_w,
                 dst_ptr, dst_row, n / 4};
    const size_t lds = (size_t)rows * C * 4;
    const dim3 grid((unsigned)((n + C - 1) / C));
    if (C == 256) hipLaunchKernelGGL(fedavg_inplace_kernel<256>, grid, dim3(256), lds, stream, a, rows, n_groups);
    else if (C == 128) hipLaunchKernelGGL(fedavg_inplace_kernel<128>, grid, dim3(256), lds, stream, a, rows, n_groups);
    else if (C == 64) hipLaunchKernelGGL(fedavg_inplace_kernel<64>, grid, dim3(256), lds, stream, a, rows, n_groups);
    else hipLaunchKernelGGL(fedavg_inplace_kernel<32>, grid, dim3(256), lds, stream, a, rows, n_groups);
    P2PFL_RETURN_LAST_ERROR();
}

P2PFL_API int p2pfl_fedavg(const float* local, long local_cs, int n_local, const float* remote,
                           long remote_cs, float* out, long out_cs, const int* src_ptr,
                           const int* src_row, const float* src_w, const int* dst_ptr,
                           const int* dst_row, int n_groups, long n, hipStream_t stream) {
    if (n_groups <= 0 || n <= 0) return -1;
    if ((n & 3) || (local_cs & 3) || (remote_cs & 3) || (out_cs & 3) || ((uintptr_t)local & 15) ||
        ((uintptr_t)remote & 15) || ((uintptr_t)out & 15))
        return -2;
    FedAvgArgs a{local, local_cs, n_local, remote, remote_cs, out, out_cs, src_ptr, src_row, src_w,
                 dst_ptr, dst_row, n / 4};
    long want = (n / 4 + 255) / 256;
    long cap = (256L * 8 + n_groups - 1) / n_groups;
    if (cap < 1) cap = 1;
    const int gx = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(fedavg_kernel, dim3(gx, n_groups), dim3(256), 0, stream, a);
    P2PFL_RETURN_LAST_ERROR();
}
