// Host data-plane helpers for the TCP compatibility path.
//
//  * PARAMS fragmenter: splits a serialized model into BLOCK_SIZE frames
//    "PARAMS" + payload ... + "\PARAMS" + zero padding, the wire format of
//    fedstellar/communication_protocol.py:737-769, in ONE pass into ONE output buffer
//    (the reference builds ~12.7k Python bytes objects for a 26 MB model).
//  * Reassembler: appends fragment payloads into a pre-sized growing buffer (amortised O(n);
//    the reference concatenates bytes objects, O(n^2): node_connection.py:341-348).
//  * Message-id ring: fixed-capacity "have I seen this gossip id" set
//    (communication_protocol.py:146-160) with O(1) lookup.
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <unordered_set>
#include <vector>

#define API extern "C" __attribute__((visibility("default")))

namespace {
const char HDR[] = "PARAMS";        // 6 bytes
const char END[] = "\\PARAMS";      // 7 bytes
constexpr size_t HDR_LEN = 6, END_LEN = 7;
}  // namespace

// Every frame is exactly `block` bytes and starts with "PARAMS"; all frames but the last carry
// block-6 payload bytes, the last carries the remaining (<= block-13, possibly 0) payload bytes,
// then "\\PARAMS", then zero padding. (The reference emits an unpadded short data frame followed
// by a 13-byte closing frame when the tail does not fit, communication_protocol.py:760-767, which
// its own fixed-BLOCK_SIZE receiver mis-frames; any reference receiver parses this layout.)
API size_t p2pfl_params_frame_count(size_t n, size_t block) {
    if (block <= HDR_LEN + END_LEN) return 0;
    const size_t cap = block - HDR_LEN, last_cap = cap - END_LEN;
    size_t frames = 1;
    while (n > last_cap) { n -= (n < cap ? n : cap); ++frames; }
    return frames;
}

// Writes the frames back to back into out (capacity frame_count*block); returns bytes written.
API size_t p2pfl_params_build(const uint8_t* data, size_t n, size_t block, uint8_t* out, size_t* last_len) {
    const size_t cap = block - HDR_LEN, last_cap = cap - END_LEN;
    size_t off = 0, w = 0;
    while (n - off > last_cap) {
        const size_t take = (n - off < cap) ? n - off : cap;
        std::memcpy(out + w, HDR, HDR_LEN);
        std::memcpy(out + w + HDR_LEN, data + off, take);
        // a tail of block-12 .. block-7 bytes cannot share a frame with the closing tag: it is zero padded
        // (< 7 bytes; the payload is a pickle, whose reader stops at its STOP opcode)
        if (take < cap) std::memset(out + w + HDR_LEN + take, 0, cap - take);
        off += take;
        w += block;
    }
    const size_t rest = n - off;
    std::memcpy(out + w, HDR, HDR_LEN);
    std::memcpy(out + w + HDR_LEN, data + off, rest);
    std::memcpy(out + w + HDR_LEN + rest, END, END_LEN);
    std::memset(out + w + HDR_LEN + rest + END_LEN, 0, block - (HDR_LEN + rest + END_LEN));
    w += block;
    *last_len = block;
    return w;
}

struct Reassembler {
    std::vector<uint8_t> buf;
};

API Reassembler* p2pfl_reasm_new(size_t reserve) {
    auto* r = new Reassembler();
    r->buf.reserve(reserve);
    return r;
}
API void p2pfl_reasm_free(Reassembler* r) { delete r; }
API void p2pfl_reasm_append(Reassembler* r, const uint8_t* p, size_t n) { r->buf.insert(r->buf.end(), p, p + n); }
API size_t p2pfl_reasm_size(Reassembler* r) { return r->buf.size(); }
API const uint8_t* p2pfl_reasm_data(Reassembler* r) { return r->buf.data(); }
API void p2pfl_reasm_clear(Reassembler* r) { r->buf.clear(); }

struct IdRing {
    size_t cap;
    std::deque<uint64_t> order;
    std::unordered_multiset<uint64_t> seen;
    std::mutex mu;
};

API IdRing* p2pfl_idring_new(size_t cap) { auto* r = new IdRing(); r->cap = cap ? cap : 1; return r; }
API void p2pfl_idring_free(IdRing* r) { delete r; }
// Returns 1 if id was already present, else inserts it (evicting the oldest) and returns 0.
API int p2pfl_idring_check_and_add(IdRing* r, uint64_t id) {
    std::lock_guard<std::mutex> g(r->mu);
    if (r->seen.count(id)) return 1;
    r->order.push_back(id);
    r->seen.insert(id);
    while (r->order.size() > r->cap) {
        auto it = r->seen.find(r->order.front());
        if (it != r->seen.end()) r->seen.erase(it);
        r->order.pop_front();
    }
    return 0;
}
API int p2pfl_idring_contains(IdRing* r, uint64_t id) {
    std::lock_guard<std::mutex> g(r->mu);
    return r->seen.count(id) ? 1 : 0;
}
API size_t p2pfl_idring_size(IdRing* r) { std::lock_guard<std::mutex> g(r->mu); return r->order.size(); }
