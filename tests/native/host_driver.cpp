// Sanitizer driver for the host runtime (ASan + UBSan, or TSan): exercises the AES link cipher, the PARAMS
// framer / reassembler at awkward sizes and the message-id ring from several threads.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {
size_t p2pfl_params_frame_count(size_t n, size_t block);
size_t p2pfl_params_build(const uint8_t* data, size_t n, size_t block, uint8_t* out, size_t* last_len);
struct Reassembler; struct IdRing;
Reassembler* p2pfl_reasm_new(size_t reserve);
void p2pfl_reasm_free(Reassembler*);
void p2pfl_reasm_append(Reassembler*, const uint8_t*, size_t);
size_t p2pfl_reasm_size(Reassembler*);
const uint8_t* p2pfl_reasm_data(Reassembler*);
void p2pfl_reasm_clear(Reassembler*);
IdRing* p2pfl_idring_new(size_t cap);
void p2pfl_idring_free(IdRing*);
int p2pfl_idring_check_and_add(IdRing*, uint64_t);
int p2pfl_idring_contains(IdRing*, uint64_t);
size_t p2pfl_idring_size(IdRing*);
void p2pfl_aes128_expand_key(const uint8_t* key16, uint8_t* out176);
int p2pfl_aes128_ecb_host(const uint8_t* key16, const uint8_t* src, uint8_t* dst, size_t n, int decrypt);
}

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    // AES: FIPS-197 appendix C.1 vector, then a round trip over a larger buffer
    const uint8_t key[16] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    const uint8_t pt[16] = {0x00, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff};
    const uint8_t ct[16] = {0x69, 0xc4, 0xe0, 0xd8, 0x6a, 0x7b, 0x04, 0x30, 0xd8, 0xcd, 0xb7, 0x80, 0x70, 0xb4, 0xc5, 0x5a};
    uint8_t out[16], rk[176];
    p2pfl_aes128_expand_key(key, rk);
    CHECK(p2pfl_aes128_ecb_host(key, pt, out, 16, 0) == 0 && std::memcmp(out, ct, 16) == 0);
    CHECK(p2pfl_aes128_ecb_host(key, ct, out, 16, 1) == 0 && std::memcmp(out, pt, 16) == 0);
    std::vector<uint8_t> big(16 * 1000 + 16), enc(big.size()), dec(big.size());
    for (size_t i = 0; i < big.size(); ++i) big[i] = (uint8_t)(i * 131 + 7);
    CHECK(p2pfl_aes128_ecb_host(key, big.data(), enc.data(), big.size(), 0) == 0);
    CHECK(p2pfl_aes128_ecb_host(key, enc.data(), dec.data(), big.size(), 1) == 0 && dec == big);

    // framing: every frame is exactly `block` bytes and nothing is written past the last one
    for (size_t block : {64u, 2048u}) {
        for (size_t n : {0u, 1u, 57u, 58u, 59u, 2041u, 2042u, 2043u, 100000u}) {
            std::vector<uint8_t> data(n + 1);
            for (size_t i = 0; i < n; ++i) data[i] = (uint8_t)(i ^ (i >> 8));
            const size_t frames = p2pfl_params_frame_count(n, block);
            std::vector<uint8_t> buf(frames * block + 1, 0xEE);
            size_t last = 0;
            CHECK(p2pfl_params_build(data.data(), n, block, buf.data(), &last) == frames * block);
            CHECK(buf[frames * block] == 0xEE);
            CHECK(frames >= 1 && std::memcmp(buf.data(), "PARAMS", 6) == 0);
        }
    }
    Reassembler* r = p2pfl_reasm_new(8);
    for (int i = 0; i < 1000; ++i) p2pfl_reasm_append(r, big.data(), 100);
    CHECK(p2pfl_reasm_size(r) == 100000 && p2pfl_reasm_data(r)[99999] == big[99]);
    p2pfl_reasm_clear(r);
    CHECK(p2pfl_reasm_size(r) == 0);
    p2pfl_reasm_free(r);

    // id ring from 4 threads: every id accepted exactly once, size bounded by the capacity
    IdRing* ring = p2pfl_idring_new(512);
    int accepted[4] = {0, 0, 0, 0};
    std::vector<std::thread> th;
    for (int t = 0; t < 4; ++t)
        th.emplace_back([&, t] { for (uint64_t i = 0; i < 400; ++i) accepted[t] += p2pfl_idring_check_and_add(ring, i) == 0; });
    for (auto& x : th) x.join();
    CHECK(accepted[0] + accepted[1] + accepted[2] + accepted[3] == 400);
    CHECK(p2pfl_idring_size(ring) <= 512 && p2pfl_idring_contains(ring, 399));
    p2pfl_idring_free(ring);
    std::printf("OK\n");
    return 0;
}
