// Host link cipher: AES-128-ECB with AES-NI (the per-link cipher of the TCP path; reference:
// fedstellar/encrypter.py:109-193 uses pycryptodome AES-ECB with a 16-byte key and space
// padding). Key expansion is exported too: the GPU kernel (csrc/aes.hip) consumes the same
// 176-byte schedule, so host and device ciphers interoperate byte for byte.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <wmmintrin.h>
#include <emmintrin.h>

#define API extern "C" __attribute__((visibility("default")))

namespace {

template <int RCON>
inline __m128i expand_step(__m128i key) {
    __m128i t = _mm_aeskeygenassist_si128(key, RCON);
    t = _mm_shuffle_epi32(t, _MM_SHUFFLE(3, 3, 3, 3));
    key = _mm_xor_si128(key, _mm_slli_si128(key, 4));
    key = _mm_xor_si128(key, _mm_slli_si128(key, 4));
    key = _mm_xor_si128(key, _mm_slli_si128(key, 4));
    return _mm_xor_si128(key, t);
}

inline void expand(const uint8_t* key, __m128i rk[11]) {
    rk[0] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(key));
    rk[1] = expand_step<0x01>(rk[0]);
    rk[2] = expand_step<0x02>(rk[1]);
    rk[3] = expand_step<0x04>(rk[2]);
    rk[4] = expand_step<0x08>(rk[3]);
    rk[5] = expand_step<0x10>(rk[4]);
    rk[6] = expand_step<0x20>(rk[5]);
    rk[7] = expand_step<0x40>(rk[6]);
    rk[8] = expand_step<0x80>(rk[7]);
    rk[9] = expand_step<0x1B>(rk[8]);
    rk[10] = expand_step<0x36>(rk[9]);
}

}  // namespace

// out[176] = the 11 AES-128 round keys.
API void p2pfl_aes128_expand_key(const uint8_t* key16, uint8_t* out176) {
    __m128i rk[11];
    expand(key16, rk);
    for (int i = 0; i < 11; ++i) _mm_storeu_si128(reinterpret_cast<__m128i*>(out176 + 16 * i), rk[i]);
}

// ECB over n bytes (n % 16 == 0). In-place allowed. Returns 0, or -1 on a bad length.
API int p2pfl_aes128_ecb_host(const uint8_t* key16, const uint8_t* src, uint8_t* dst, size_t n, int decrypt) {
    if (n % 16) return -1;
    __m128i rk[11];
    expand(key16, rk);
    if (!decrypt) {
        for (size_t i = 0; i < n; i += 16) {
            __m128i s = _mm_xor_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), rk[0]);
            for (int r = 1; r < 10; ++r) s = _mm_aesenc_si128(s, rk[r]);
            s = _mm_aesenclast_si128(s, rk[10]);
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), s);
        }
    } else {
        __m128i dk[11];
        dk[0] = rk[10];
        for (int r = 1; r < 10; ++r) dk[r] = _mm_aesimc_si128(rk[10 - r]);
        dk[10] = rk[0];
        for (size_t i = 0; i < n; i += 16) {
            __m128i s = _mm_xor_si128(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), dk[0]);
            for (int r = 1; r < 10; ++r) s = _mm_aesdec_si128(s, dk[r]);
            s = _mm_aesdeclast_si128(s, dk[10]);
            _mm_storeu_si128(reinterpret_cast<__m128i*>(dst + i), s);
        }
    }
    return 0;
}
