// AES-128-ECB over flat parameter rows on the GPU (encrypted model exchange, BASELINE config #4).
//
// The reference encrypts every 2 KiB TCP fragment of a model with AES-ECB on the host
// (fedstellar/encrypter.py:121-145, node_connection.py:382-386). Here the flat fp32 row is
// encrypted in device memory right before it leaves over xGMI/TCP and decrypted on arrival:
// one 16-byte block per lane, T-table rounds out of LDS (tables are rebuilt per workgroup from
// the 256-byte S-box; 1 KiB each). Key expansion (11 round keys = 176 bytes) is done on the host.
#include "common.h"

namespace {

__device__ const uint8_t SBOX[256] = {
    0x63,0x7c,0x77,0x7b,0xf2,0x6b,0x6f,0xc5,0x30,0x01,0x67,0x2b,0xfe,0xd7,0xab,0x76,
    0xca,0x82,0xc9,0x7d,0xfa,0x59,0x47,0xf0,0xad,0xd4,0xa2,0xaf,0x9c,0xa4,0x72,0xc0,
    0xb7,0xfd,0x93,0x26,0x36,0x3f,0xf7,0xcc,0x34,0xa5,0xe5,0xf1,0x71,0xd8,0x31,0x15,
    0x04,0xc7,0x23,0xc3,0x18,0x96,0x05,0x9a,0x07,0x12,0x80,0xe2,0xeb,0x27,0xb2,0x75,
    0x09,0x83,0x2c,0x1a,0x1b,0x6e,0x5a,0xa0,0x52,0x3b,0xd6,0xb3,0x29,0xe3,0x2f,0x84,
    0x53,0xd1,0x00,0xed,0x20,0xfc,0xb1,0x5b,0x6a,0xcb,0xbe,0x39,0x4a,0x4c,0x58,0xcf,
    0xd0,0xef,0xaa,0xfb,0x43,0x4d,0x33,0x85,0x45,0xf9,0x02,0x7f,0x50,0x3c,0x9f,0xa8,
    0x51,0xa3,0x40,0x8f,0x92,0x9d,0x38,0xf5,0xbc,0xb6,0xda,0x21,0x10,0xff,0xf3,0xd2,
    0xcd,0x0c,0x13,0xec,0x5f,0x97,0x44,0x17,0xc4,0xa7,0x7e,0x3d,0x64,0x5d,0x19,0x73,
    0x60,0x81,0x4f,0xdc,0x22,0x2a,0x90,0x88,0x46,0xee,0xb8,0x14,0xde,0x5e,0x0b,0xdb,
    0xe0,0x32,0x3a,0x0a,0x49,0x06,0x24,0x5c,0xc2,0xd3,0xac,0x62,0x91,0x95,0xe4,0x79,
    0xe7,0xc8,0x37,0x6d,0x8d,0xd5,0x4e,0xa9,0x6c,0x56,0xf4,0xea,0x65,0x7a,0xae,0x08,
    0xba,0x78,0x25,0x2e,0x1c,0xa6,0xb4,0xc6,0xe8,0xdd,0x74,0x1f,0x4b,0xbd,0x8b,0x8a,
    0x70,0x3e,0xb5,0x66,0x48,0x03,0xf6,0x0e,0x61,0x35,0x57,0xb9,0x86,0xc1,0x1d,0x9e,
    0xe1,0xf8,0x98,0x11,0x69,0xd9,0x8e,0x94,0x9b,0x1e,0x87,0xe9,0xce,0x55,0x28,0xdf,
    0x8c,0xa1,0x89,0x0d,0xbf,0xe6,0x42,0x68,0x41,0x99,0x2d,0x0f,0xb0,0x54,0xbb,0x16};

__device__ __forceinline__ uint32_t xt(uint32_t b) { return ((b << 1) ^ ((b & 0x80) ? 0x1b : 0)) & 0xff; }
__device__ __forceinline__ uint32_t gmul(uint32_t b, int m) {   // m in {9, 11, 13, 14}
    const uint32_t b2 = xt(b), b4 = xt(b2), b8 = xt(b4);
    uint32_t r = 0;
    if (m & 1) r ^= b;
    if (m & 2) r ^= b2;
    if (m & 4) r ^= b4;
    if (m & 8) r ^= b8;
    return r;
}
__device__ __forceinline__ uint32_t rotl(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
__device__ __forceinline__ uint32_t inv_mix_word(uint32_t w) {
    const uint32_t s0 = w & 0xff, s1 = (w >> 8) & 0xff, s2 = (w >> 16) & 0xff, s3 = w >> 24;
    const uint32_t o0 = gmul(s0, 14) ^ gmul(s1, 11) ^ gmul(s2, 13) ^ gmul(s3, 9);
    const uint32_t o1 = gmul(s0, 9) ^ gmul(s1, 14) ^ gmul(s2, 11) ^ gmul(s3, 13);
    const uint32_t o2 = gmul(s0, 13) ^ gmul(s1, 9) ^ gmul(s2, 14) ^ gmul(s3, 11);
    const uint32_t o3 = gmul(s0, 11) ^ gmul(s1, 13) ^ gmul(s2, 9) ^ gmul(s3, 14);
    return o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}

template <bool DEC>
__global__ __launch_bounds__(256) void aes128_ecb_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                         const uint8_t* __restrict__ rk_bytes, long nblocks) {
    __shared__ uint32_t T[256];     // Te0 (encrypt) or Td0 (decrypt); other columns are rotations
    __shared__ uint8_t S[256];      // S-box (encrypt) or inverse S-box (decrypt) for the last round
    __shared__ uint32_t rk[44];
    const int t = threadIdx.x;
    if (DEC) S[SBOX[t]] = (uint8_t)t; else S[t] = SBOX[t];
    if (t < 44) {
        const uint8_t* p = rk_bytes + 4 * t;
        rk[t] = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    __syncthreads();
    {
        const uint32_t s = S[t];
        if (DEC) T[t] = gmul(s, 14) | (gmul(s, 9) << 8) | (gmul(s, 13) << 16) | (gmul(s, 11) << 24);
        else     T[t] = xt(s) | (s << 8) | (s << 16) | ((xt(s) ^ s) << 24);
    }
    if (DEC && t >= 4 && t < 40) rk[t] = inv_mix_word(rk[t]);   // equivalent inverse cipher keys
    __syncthreads();

    for (long i = (long)blockIdx.x * 256 + t; i < nblocks; i += (long)gridDim.x * 256) {
        const uint4 in = src[i];
        uint32_t s0, s1, s2, s3;
        if (!DEC) {
            s0 = in.x ^ rk[0]; s1 = in.y ^ rk[1]; s2 = in.z ^ rk[2]; s3 = in.w ^ rk[3];
#pragma unroll
            for (int r = 1; r < 10; ++r) {
                const uint32_t t0 = T[s0 & 0xff] ^ rotl(T[(s1 >> 8) & 0xff], 8) ^ rotl(T[(s2 >> 16) & 0xff], 16) ^ rotl(T[s3 >> 24], 24) ^ rk[4 * r];
                const uint32_t t1 = T[s1 & 0xff] ^ rotl(T[(s2 >> 8) & 0xff], 8) ^ rotl(T[(s3 >> 16) & 0xff], 16) ^ rotl(T[s0 >> 24], 24) ^ rk[4 * r + 1];
                const uint32_t t2 = T[s2 & 0xff] ^ rotl(T[(s3 >> 8) & 0xff], 8) ^ rotl(T[(s0 >> 16) & 0xff], 16) ^ rotl(T[s1 >> 24], 24) ^ rk[4 * r + 2];
                const uint32_t t3 = T[s3 & 0xff] ^ rotl(T[(s0 >> 8) & 0xff], 8) ^ rotl(T[(s1 >> 16) & 0xff], 16) ^ rotl(T[s2 >> 24], 24) ^ rk[4 * r + 3];
                s0 = t0; s1 = t1; s2 = t2; s3 = t3;
            }
            const uint32_t o0 = (S[s0 & 0xff] | (S[(s1 >> 8) & 0xff] << 8) | (S[(s2 >> 16) & 0xff] << 16) | ((uint32_t)S[s3 >> 24] << 24)) ^ rk[40];
            const uint32_t o1 = (S[s1 & 0xff] | (S[(s2 >> 8) & 0xff] << 8) | (S[(s3 >> 16) & 0xff] << 16) | ((uint32_t)S[s0 >> 24] << 24)) ^ rk[41];
            const uint32_t o2 = (S[s2 & 0xff] | (S[(s3 >> 8) & 0xff] << 8) | (S[(s0 >> 16) & 0xff] << 16) | ((uint32_t)S[s1 >> 24] << 24)) ^ rk[42];
            const uint32_t o3 = (S[s3 & 0xff] | (S[(s0 >> 8) & 0xff] << 8) | (S[(s1 >> 16) & 0xff] << 16) | ((uint32_t)S[s2 >> 24] << 24)) ^ rk[43];
            dst[i] = make_uint4(o0, o1, o2, o3);
        } else {
            s0 = in.x ^ rk[40]; s1 = in.y ^ rk[41]; s2 = in.z ^ rk[42]; s3 = in.w ^ rk[43];
#pragma unroll
            for (int r = 9; r >= 1; --r) {
                const uint32_t t0 = T[s0 & 0xff] ^ rotl(T[(s3 >> 8) & 0xff], 8) ^ rotl(T[(s2 >> 16) & 0xff], 16) ^ rotl(T[s1 >> 24], 24) ^ rk[4 * r];
                const uint32_t t1 = T[s1 & 0xff] ^ rotl(T[(s0 >> 8) & 0xff], 8) ^ rotl(T[(s3 >> 16) & 0xff], 16) ^ rotl(T[s2 >> 24], 24) ^ rk[4 * r + 1];
                const uint32_t t2 = T[s2 & 0xff] ^ rotl(T[(s1 >> 8) & 0xff], 8) ^ rotl(T[(s0 >> 16) & 0xff], 16) ^ rotl(T[s3 >> 24], 24) ^ rk[4 * r + 2];
                const uint32_t t3 = T[s3 & 0xff] ^ rotl(T[(s2 >> 8) & 0xff], 8) ^ rotl(T[(s1 >> 16) & 0xff], 16) ^ rotl(T[s0 >> 24], 24) ^ rk[4 * r + 3];
                s0 = t0; s1 = t1; s2 = t2; s3 = t3;
            }
            const uint32_t o0 = (S[s0 & 0xff] | (S[(s3 >> 8) & 0xff] << 8) | (S[(s2 >> 16) & 0xff] << 16) | ((uint32_t)S[s1 >> 24] << 24)) ^ rk[0];
            const uint32_t o1 = (S[s1 & 0xff] | (S[(s0 >> 8) & 0xff] << 8) | (S[(s3 >> 16) & 0xff] << 16) | ((uint32_t)S[s2 >> 24] << 24)) ^ rk[1];
            const uint32_t o2 = (S[s2 & 0xff] | (S[(s1 >> 8) & 0xff] << 8) | (S[(s0 >> 16) & 0xff] << 16) | ((uint32_t)S[s3 >> 24] << 24)) ^ rk[2];
            const uint32_t o3 = (S[s3 & 0xff] | (S[(s2 >> 8) & 0xff] << 8) | (S[(s1 >> 16) & 0xff] << 16) | ((uint32_t)S[s0 >> 24] << 24)) ^ rk[3];
            dst[i] = make_uint4(o0, o1, o2, o3);
        }
    }
}

}  // namespace

P2PFL_API int p2pfl_aes128_ecb(const void* src, void* dst, const uint8_t* round_keys, long nblocks,
                               int decrypt, hipStream_t stream) {
    if (nblocks <= 0) return -1;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return -2;
    long want = (nblocks + 255) / 256;
    const int gx = (int)(want < 2048 ? want : 2048);
    if (decrypt)
        hipLaunchKernelGGL(aes128_ecb_kernel<true>, dim3(gx), dim3(256), 0, stream, (const uint4*)src,
                           (uint4*)dst, round_keys, nblocks);
    else
        hipLaunchKernelGGL(aes128_ecb_kernel<false>, dim3(gx), dim3(256), 0, stream, (const uint4*)src,
                           (uint4*)dst, round_keys, nblocks);
    P2PFL_RETURN_LAST_ERROR();
}
