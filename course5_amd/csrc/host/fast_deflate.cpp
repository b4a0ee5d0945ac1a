// A deflate encoder for ONE kind of input: doubles that were widened from floats (the .vti image the reference writes:
// object2d.cpp:13,17-21 allocates VTK_DOUBLE, plane.cpp:165-166 fills it from floats).  Such a double's three low
// bytes are zero and its fourth has five zero bits, so the parse is known in advance and needs no match search:
//     first value        8 literals
//     value == previous  joins a run: one match (distance 8) per 256 bytes of run
//     any other value    match (length 3, distance 8) for the three zero bytes + 5 literals
// with one dynamic Huffman block per call, its code lengths from the block's own histogram (the exponent bytes take 2-4
// bits, the mantissa bytes stay near 8).  The output is an ordinary zlib stream (RFC 1950 / 1951): any inflate reads it
// (vtkZLibDataCompressor, Python's zlib in the tests).  zlib's own level 1 spends ~8 ns per input byte looking for
// matches this input does not have; this spends about one.
#include "fast_deflate.hpp"

#include <algorithm>
#include <cstring>

namespace c5 {
namespace {

struct LengthCode {
    uint16_t code;
    uint8_t extra_bits;
    uint16_t extra;
};

// RFC 1951, 3.2.5
LengthCode length_code(unsigned len) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t bits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    int k = 28;
    while (k > 0 && base[k] > len) --k;
    return LengthCode{static_cast<uint16_t>(257 + k), bits[k], static_cast<uint16_t>(len - base[k])};
}

// Huffman code lengths of at most max_bits for the symbols with freq > 0 (others 0).  A tree deeper than max_bits is
// rebuilt from halved frequencies (none dropping to zero) until it fits: not optimal, always valid.
void huffman_lengths(const uint32_t* freq_in, int n, int max_bits, uint8_t* len) {
    uint32_t freq[288];
    for (int i = 0; i < n; ++i) freq[i] = freq_in[i];
    for (;;) {
        int sym[288], m = 0;
        for (int i = 0; i < n; ++i) {
            len[i] = 0;
            if (freq[i]) sym[m++] = i;
        }
        if (m == 0) return;
        if (m == 1) {
            len[sym[0]] = 1;
            return;
        }
        // nodes 0..m-1 leaves, m.. internal; two queues (leaves sorted by weight, internal nodes in creation order)
        std::sort(sym, sym + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        uint64_t weight[2 * 288];
        int parent[2 * 288];
        for (int i = 0; i < m; ++i) weight[i] = freq[sym[i]];
        int leaf = 0, inner = m, made = m;
        auto take = [&]() {
            if (leaf < m && (inner >= made || weight[leaf] <= weight[inner])) return leaf++;
            return inner++;
        };
        while (made < 2 * m - 1) {
            const int a = take(), b = take();
            weight[made] = weight[a] + weight[b];
            parent[a] = parent[b] = made;
            ++made;
        }
        int deepest = 0;
        int depth[2 * 288];
        depth[2 * m - 2] = 0;
        for (int i = 2 * m - 3; i >= 0; --i) {
            depth[i] = depth[parent[i]] + 1;
            if (i < m && depth[i] > deepest) deepest = depth[i];
        }
        if (deepest <= max_bits) {
            for (int i = 0; i < m; ++i) len[sym[i]] = static_cast<uint8_t>(depth[i]);
            return;
        }
        for (int i = 0; i < n; ++i)
            if (freq[i]) freq[i] = (freq[i] + 1) / 2;
    }
}

// canonical codes (RFC 1951, 3.2.2), bit-reversed: deflate packs Huffman codes starting from their most significant bit
void canonical_codes(const uint8_t* len, int n, uint16_t* code) {
    unsigned count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    unsigned c = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + count[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) {
            code[i] = 0;
            continue;
        }
        unsigned v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        code[i] = static_cast<uint16_t>(r);
    }
}

struct BitWriter {
    unsigned char* p;
    unsigned char* end;
    uint64_t acc = 0;
    int n = 0;
    bool ok = true;
    void put(uint32_t v, int bits) {  // bits <= 32
        acc |= static_cast<uint64_t>(v) << n;
        n += bits;
        if (n >= 32) {
            if (end - p < 4) {
                ok = false;
                n = 0;
                acc = 0;
                return;
            }
            const uint32_t w = static_cast<uint32_t>(acc);
            std::memcpy(p, &w, 4);  // little endian host (the file format says so too)
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    void finish() {
        while (n > 0) {
            if (p == end) {
                ok = false;
                return;
            }
            *p++ = static_cast<unsigned char>(acc);
            acc >>= 8;
            n -= 8;
        }
        n = 0;
    }
};

// word(i): the i-th value's eight bytes as a little-endian word (byte k = bits 8k ... 8k + 7)
template <class Word>
size_t encode(Word word, size_t count, unsigned char* out, size_t cap) {
    if (count < 2 || count > (size_t{1} << 24) || cap < 64) return 0;
    auto byte_of = [](uint64_t w, int b) { return static_cast<unsigned>((w >> (8 * b)) & 0xFFu); };
    // pass 1: is it what it should be, how often does every symbol occur, and the stream's Adler-32 (RFC 1950: of the
    // uncompressed bytes; the three zero bytes of a value add nothing to s1 and 3 s1 to s2)
    uint32_t f_lit[288] = {0}, f_dist[30] = {0};
    uint64_t s1 = 1, s2 = 0;  // (no overflow below 2^24 values: s1 < 2^35, s2 < 2^62)
    auto adler_value = [&](uint64_t w) {
        s2 += 3 * s1;
        for (int b = 3; b < 8; ++b) {
            s1 += byte_of(w, b);
            s2 += s1;
        }
    };
    const LengthCode run_full = length_code(256), three = length_code(3);
    {
        uint64_t prev = word(0);
        if (prev & 0xFFFFFFull) return 0;
        for (int b = 0; b < 8; ++b) ++f_lit[byte_of(prev, b)];
        adler_value(prev);
        size_t run = 0;
        auto flush = [&]() {
            size_t left = 8 * run;
            if (left >= 256) {
                f_lit[run_full.code] += static_cast<uint32_t>(left / 256);
                f_dist[5] += static_cast<uint32_t>(left / 256);
                left %= 256;
            }
            if (left) {
                ++f_lit[length_code(static_cast<unsigned>(left)).code];
                ++f_dist[5];
            }
            run = 0;
        };
        for (size_t i = 1; i < count; ++i) {
            const uint64_t w = word(i);
            if (w & 0xFFFFFFull) return 0;  // not a widened float: the caller's general compressor takes it
            adler_value(w);
            if (w == prev) {
                ++run;
                continue;
            }
            if (run) flush();
            ++f_lit[three.code];
            ++f_dist[5];
            for (int b = 3; b < 8; ++b) ++f_lit[byte_of(w, b)];
            prev = w;
        }
        if (run) flush();
        f_lit[256] = 1;
    }
    // code lengths and codes
    uint8_t l_lit[288], l_dist[30] = {0};
    uint16_t c_lit[288], c_dist[30] = {0};
    huffman_lengths(f_lit, 286, 15, l_lit);
    l_lit[286] = l_lit[287] = 0;
    canonical_codes(l_lit, 286, c_lit);
    l_dist[5] = 1;  // the one distance there is: code 5 (distances 7-8), one bit (RFC 1951, 3.2.7: a single code of one bit)
    c_dist[5] = 0;
    int hlit = 286;
    while (hlit > 257 && l_lit[hlit - 1] == 0) --hlit;
    const int hdist = 6;
    // the code lengths, runs of zeros folded (symbols 17 and 18), as symbols of the code length alphabet
    uint8_t seq[320], seq_extra[320];
    int n_seq = 0;
    {
        uint8_t all[316];
        int n_all = 0;
        for (int i = 0; i < hlit; ++i) all[n_all++] = l_lit[i];
        for (int i = 0; i < hdist; ++i) all[n_all++] = l_dist[i];
        for (int i = 0; i < n_all;) {
            if (all[i] != 0) {
                seq[n_seq] = all[i];
                seq_extra[n_seq++] = 0;
                ++i;
                continue;
            }
            int z = 1;
            while (i + z < n_all && all[i + z] == 0) ++z;
            i += z;
            while (z > 0) {
                if (z >= 11) {
                    const int t = std::min(z, 138);
                    seq[n_seq] = 18;
                    seq_extra[n_seq++] = static_cast<uint8_t>(t - 11);
                    z -= t;
                } else if (z >= 3) {
                    seq[n_seq] = 17;
                    seq_extra[n_seq++] = static_cast<uint8_t>(z - 3);
                    z = 0;
                } else {
                    seq[n_seq] = 0;
                    seq_extra[n_seq++] = 0;
                    --z;
                }
            }
        }
    }
    uint32_t f_cl[19] = {0};
    for (int i = 0; i < n_seq; ++i) ++f_cl[seq[i]];
    uint8_t l_cl[19];
    uint16_t c_cl[19];
    huffman_lengths(f_cl, 19, 7, l_cl);
    canonical_codes(l_cl, 19, c_cl);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && l_cl[order[hclen - 1]] == 0) --hclen;

    // the stream
    out[0] = 0x78;  // deflate, 32 KB window
    out[1] = 0x01;  // (0x7801 is a multiple of 31: no preset dictionary, "fastest")
    BitWriter w{out + 2, out + cap - 4};
    w.put(1, 1);  // last block
    w.put(2, 2);  // dynamic Huffman codes
    w.put(static_cast<uint32_t>(hlit - 257), 5);
    w.put(static_cast<uint32_t>(hdist - 1), 5);
    w.put(static_cast<uint32_t>(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) w.put(l_cl[order[i]], 3);
    for (int i = 0; i < n_seq; ++i) {
        w.put(c_cl[seq[i]], l_cl[seq[i]]);
        if (seq[i] == 17) w.put(seq_extra[i], 3);
        if (seq[i] == 18) w.put(seq_extra[i], 7);
    }
    auto put_match = [&](const LengthCode& lc) {
        w.put(c_lit[lc.code], l_lit[lc.code]);
        if (lc.extra_bits) w.put(lc.extra, lc.extra_bits);
        w.put(c_dist[5], 1);
        w.put(1, 1);  // distance 8 = base 7 + 1
    };
    {
        uint64_t prev = word(0);
        for (int b = 0; b < 8; ++b) w.put(c_lit[byte_of(prev, b)], l_lit[byte_of(prev, b)]);
        size_t run = 0;
        auto flush = [&]() {
            size_t left = 8 * run;
            for (; left >= 256; left -= 256) put_match(run_full);
            if (left) put_match(length_code(static_cast<unsigned>(left)));
            run = 0;
        };
        for (size_t i = 1; i < count && w.ok; ++i) {
            const uint64_t v = word(i);
            if (v == prev) {
                ++run;
                continue;
            }
            if (run) flush();
            // the three zero bytes as a match, the five others as literals: at most 13 + 5 x 15 bits, in three puts
            const unsigned q[8] = {0, 0, 0, byte_of(v, 3), byte_of(v, 4), byte_of(v, 5), byte_of(v, 6), byte_of(v, 7)};
            w.put(c_lit[three.code], l_lit[three.code]);
            w.put(static_cast<uint32_t>(c_dist[5]) | 2u, 2);  // distance code (one bit: 0) + extra bit 1
            w.put(static_cast<uint32_t>(c_lit[q[3]]) | (static_cast<uint32_t>(c_lit[q[4]]) << l_lit[q[3]]), l_lit[q[3]] + l_lit[q[4]]);
            w.put(static_cast<uint32_t>(c_lit[q[5]]) | (static_cast<uint32_t>(c_lit[q[6]]) << l_lit[q[5]]), l_lit[q[5]] + l_lit[q[6]]);
            w.put(c_lit[q[7]], l_lit[q[7]]);
            prev = v;
        }
        if (run) flush();
    }
    w.put(c_lit[256], l_lit[256]);
    w.finish();
    if (!w.ok) return 0;
    const uint32_t adler = (static_cast<uint32_t>(s2 % 65521u) << 16) | static_cast<uint32_t>(s1 % 65521u);
    w.p[0] = static_cast<unsigned char>(adler >> 24);
    w.p[1] = static_cast<unsigned char>(adler >> 16);
    w.p[2] = static_cast<unsigned char>(adler >> 8);
    w.p[3] = static_cast<unsigned char>(adler);
    return static_cast<size_t>(w.p + 4 - out);
}

}  // namespace

size_t deflate_widened_doubles(const double* vals, size_t count, unsigned char* out, size_t cap) {
    const unsigned char* bytes = reinterpret_cast<const unsigned char*>(vals);
    return encode(
        [bytes](size_t i) {
            uint64_t w;
            std::memcpy(&w, bytes + 8 * i, 8);
            return w;
        },
        count, out, cap);
}

size_t deflate_floats_as_doubles(const float* vals, size_t count, unsigned char* out, size_t cap) {
    return encode(
        [vals](size_t i) {
            const double d = static_cast<double>(vals[i]);  // (exact: every float is a double)
            uint64_t w;
            std::memcpy(&w, &d, 8);
            return w;
        },
        count, out, cap);
}

}  // namespace c5
