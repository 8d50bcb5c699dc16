// One-shot gzip (RFC 1952 / 1951) decoder for the input side of the deploy loops: nib.load(...).get_data() of the reference
// (common/deploy_network.py:80-83, deploy_network_ao.py:88-92) inflates a 40 MB cine per short-axis subject, and with the network
// at 11 ms per subject that inflate -- 0.26 s of a core through zlib 1.2.11 (~145 MB/s on literal-heavy int16 MR data) -- is what
// bounds a cohort run.  This decoder does the same job at several times zlib's rate on the same data:
//   * whole input and whole output in memory (the reader already holds the file; the destination is the staging buffer), so there
//     is no window, no resumable state machine, and matches copy straight from the output;
//   * a 64-bit bit buffer refilled without a branch (one unaligned 8-byte load), up to three literals per refill;
//   * an 11-bit first-level table for literal / length codes and an 8-bit one for distances, longer codes through second-level
//     tables; an entry carries the symbol's base value, its extra-bit count and the code length, so a symbol costs one lookup;
//   * matches copied 16 bytes at a time (8 for distances 8..15, a broadcast store for distance 1);
//   * CRC-32 by carry-less multiplication where the CPU has it, slicing-by-8 otherwise.
// It is strict: anything it does not like (header CRC flag, reserved bits, invalid codes, a distance beyond the output written so
// far, bad CRC-32 / ISIZE, trailing garbage) returns an error code and the Python reader falls back to zlib, which then raises --
// or accepts -- exactly as before.  No code here is on the GPU path; built into libukbb_labelgz.so (host only) and libukbb_fcn.so.
#include <cstdint>
#include <cstring>
#include <cstddef>

#include "../../include/ukbb_fcn.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

#ifndef UKBB_GZ_LL_BITS
#define UKBB_GZ_LL_BITS 11
#endif
constexpr int LL_BITS = UKBB_GZ_LL_BITS, D_BITS = 8;   // first-level table widths
constexpr int LL_SIZE = (1 << LL_BITS) + 288 * 16;      // + every possible second-level entry (codes of up to 15 bits)
constexpr int D_SIZE = (1 << D_BITS) + 32 * 128;

// Table entry: bits 0-7 bits to drop for this lookup (code bits; for length / distance symbols code + extra bits), bits 8-11
// extra-bit count (literal entries: number of literals, 1 or 2; second-level pointer: index bits of that table), bits 12-15 kind,
// bits 16-31 base value (literal byte(s), length base, distance base, or second-level table start).
constexpr uint32_t K_LIT = 0x1000, K_SUB = 0x2000, K_EOB = 0x4000, K_BAD = 0x8000;
inline uint32_t mk(uint32_t base, uint32_t extra, uint32_t len, uint32_t kind) { return (base << 16) | kind | (extra << 8) | len; }

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t bitrev(uint32_t v, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// Canonical Huffman code of `n` symbols with lengths lens[] (0 = unused, <= 15) -> two-level decode table.
// kind: 0 literal / length alphabet, 1 distance alphabet.  Returns 0, or -1 for an over-subscribed or (other than the one-code
// distance set RFC 1951 allows) incomplete code.
int build_table(const uint8_t *lens, int n, int kind, uint32_t *tab, int root) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    int used = n - count[0];
    long left = 1;                                          // Kraft sum bookkeeping, as a count of unassigned codes
    for (int l = 1; l <= 15; ++l) { left = (left << 1) - count[l]; if (left < 0) return -1; }
    bool incomplete = left > 0;
    // only "at most one distance code" may be incomplete, and (as zlib's inflate_table demands) that one code has length 1: a
    // longer single code would leave second-level slots unwritten
    if (incomplete && !(kind == 1 && (used == 0 || (used == 1 && count[1] == 1)))) return -1;
    const int nroot = 1 << root;
    if (incomplete || used == 0) for (int i = 0; i < nroot; ++i) tab[i] = mk(0, 0, 1, K_BAD);
    if (used == 0) return 0;                                // block with literals only: any distance code is an error when met
    uint32_t next[16]; next[0] = 0; next[1] = 0;
    for (int l = 1; l < 15; ++l) next[l + 1] = (next[l] + count[l]) << 1;
    auto entry = [&](int sym, int drop) -> uint32_t {
        // length / distance symbols: bits 0-5 = code bits + extra bits (dropped in one shift), bits 8-11 = the extra-bit count
        if (kind == 1) return sym < 30 ? mk(DIST_BASE[sym], DIST_EXTRA[sym], drop + DIST_EXTRA[sym], 0) : mk(0, 0, drop, K_BAD);
        if (sym < 256) return mk(sym, 1, drop, K_LIT);
        if (sym == 256) return mk(0, 0, drop, K_EOB);
        return sym < 286 ? mk(LEN_BASE[sym - 257], LEN_EXTRA[sym - 257], drop + LEN_EXTRA[sym - 257], 0) : mk(0, 0, drop, K_BAD);
    };
    // second-level tables: one per root-bit prefix that long codes share, sized by the longest code under that prefix
    uint8_t sub_bits[1 << LL_BITS];
    bool any_long = false;
    for (int l = root + 1; l <= 15; ++l) if (count[l]) any_long = true;
    uint32_t codes[320];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        codes[s] = l ? bitrev(next[l]++, l) : 0;
    }
    if (any_long) {
        memset(sub_bits, 0, nroot);
        for (int s = 0; s < n; ++s) {
            const int l = lens[s];
            if (l > root) { const uint32_t p = codes[s] & (nroot - 1); if (l - root > sub_bits[p]) sub_bits[p] = (uint8_t)(l - root); }
        }
        uint32_t base = nroot;
        for (int p = 0; p < nroot; ++p)
            if (sub_bits[p]) {
                tab[p] = mk(base, sub_bits[p], root, K_SUB);
                // an incomplete code cannot reach here (the one allowed has length 1 <= root), so every second-level slot gets filled below
                base += 1u << sub_bits[p];
            }
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        if (l <= root) {
            const uint32_t e = entry(s, l);
            for (uint32_t i = codes[s]; i < (uint32_t)nroot; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t p = codes[s] & (nroot - 1);
            const uint32_t b = tab[p] >> 16, sb = sub_bits[p];
            const uint32_t e = entry(s, l - root);
            for (uint32_t i = codes[s] >> root; i < (1u << sb); i += 1u << (l - root)) tab[b + i] = e;
        }
    }
    if (kind == 0) {
        // Two literals per lookup where both codes fit the first-level index (int16 MR data: a high byte of 2-4 bits next to a low
        // byte of 7-9): the entry holds both bytes and the sum of the code lengths, the decoder stores two bytes and advances by the
        // entry's count.  The chain index -> load -> shift is what bounds a literal-heavy stream, so this halves it where it applies.
        uint32_t single[1 << LL_BITS];
        memcpy(single, tab, sizeof(single));
        for (int i = 0; i < nroot; ++i) {
            const uint32_t e1 = single[i];
            if (!(e1 & K_LIT)) continue;
            const int l1 = e1 & 0xff;
            if (l1 >= root) continue;
            const uint32_t e2 = single[i >> l1];                 // unknown upper bits read as 0: valid iff the second code fits the known ones
            if (!(e2 & K_LIT)) continue;
            const int l2 = e2 & 0xff;
            if (l1 + l2 > root) continue;
            tab[i] = mk((e1 >> 16) | ((e2 >> 16) << 8), 2, l1 + l2, K_LIT);
        }
    }
    return 0;
}

struct Tables {
    uint32_t ll[LL_SIZE];
    uint32_t d[D_SIZE];
};

struct Fixed {
    Tables t;
    Fixed() {
        uint8_t l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, 0, t.ll, LL_BITS);
        uint8_t d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, 1, t.d, D_BITS);
    }
};
const Tables &fixed_tables() { static const Fixed f; return f.t; }

inline uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }       // little-endian hosts only (x86-64)
inline void copy8(uint8_t *d, const uint8_t *s) { uint64_t v; memcpy(&v, s, 8); memcpy(d, &v, 8); }
inline void copy16(uint8_t *d, const uint8_t *s) {
#if defined(__x86_64__)
    _mm_storeu_si128((__m128i *)d, _mm_loadu_si128((const __m128i *)s));
#else
    copy8(d, s); copy8(d + 8, s + 8);
#endif
}

enum { E_INPUT = -1, E_DATA = -2, E_OUTPUT = -3 };        // ran out of input / invalid stream / does not fit the output buffer

struct Bits {
    const uint8_t *in, *end;
    uint64_t buf = 0;
    int cnt = 0;
    int over = 0;                                           // bytes of zero padding shifted in beyond the end of the input
    // careful refill: byte by byte, zeros beyond the end (callers check `over` against what they consumed)
    inline void fill_safe() {
        while (cnt <= 56) {
            if (in < end) buf |= (uint64_t)*in++ << cnt; else ++over;
            cnt += 8;
        }
    }
    inline uint32_t take(int n) { const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1)); buf >>= n; cnt -= n; return v; }
    inline bool overrun() const { return over * 8 > cnt; }   // consumed bits that were never in the input
};

// One deflate stream from b (positioned at its first block header) into out[pos..cap).  Returns the new position or an error.
int64_t inflate_stream(Bits &b, uint8_t *const out, uint64_t pos, const uint64_t cap) {
    static thread_local Tables dyn;
    bool last;
    do {
        b.fill_safe();
        last = b.take(1);
        const uint32_t type = b.take(2);
        if (b.overrun()) return E_INPUT;
        if (type == 0) {                                    // stored: drop to a byte boundary, LEN / NLEN, raw bytes
            b.take(b.cnt & 7);
            // hand the whole bytes still in the bit buffer back to the input
            const int back = b.cnt >> 3;
            const int real = back - b.over > 0 ? back - b.over : 0;
            b.in -= real; b.buf = 0; b.cnt = 0; b.over = 0;
            if (b.end - b.in < 4) return E_INPUT;
            const uint32_t len = b.in[0] | (b.in[1] << 8), nlen = b.in[2] | (b.in[3] << 8);
            if ((len ^ nlen) != 0xffff) return E_DATA;
            b.in += 4;
            if ((uint64_t)(b.end - b.in) < len) return E_INPUT;
            if (cap - pos < len) return E_OUTPUT;
            memcpy(out + pos, b.in, len);
            b.in += len; pos += len;
            continue;
        }
        if (type == 3) return E_DATA;
        const Tables *t;
        if (type == 1) {
            t = &fixed_tables();
        } else {
            b.fill_safe();
            const int hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
            if (hlit > 286 || hdist > 30) return E_DATA;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) { if (b.cnt < 3) b.fill_safe(); cl[order[i]] = (uint8_t)b.take(3); }
            uint32_t pre[1 << 7];
            {   // code-length code: <= 7 bits, one level; reuse the builder with the literal alphabet and read only `base`
                uint8_t tmp[19];
                memcpy(tmp, cl, 19);
                int count[8] = {0};
                for (int i = 0; i < 19; ++i) ++count[tmp[i]];
                long left = 1;
                for (int l = 1; l <= 7; ++l) { left = (left << 1) - count[l]; if (left < 0) return E_DATA; }
                if (left > 0) return E_DATA;                    // zlib rejects an incomplete code-length code too (unless a single code: rare, let zlib decide)
                uint32_t next[9]; next[1] = 0;
                for (int l = 1; l < 8; ++l) next[l + 1] = (next[l] + count[l]) << 1;
                for (int s = 0; s < 19; ++s) {
                    const int l = tmp[s];
                    if (!l) continue;
                    const uint32_t c = bitrev(next[l]++, l);
                    for (uint32_t i = c; i < 128; i += 1u << l) pre[i] = (uint32_t)(s << 8) | l;
                }
            }
            uint8_t lens[320];
            int i = 0;
            const int total = hlit + hdist;
            while (i < total) {
                b.fill_safe();
                const uint32_t e = pre[b.buf & 127];
                b.take(e & 0xff);
                const int sym = e >> 8;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep; uint8_t v = 0;
                if (sym == 16) { if (i == 0) return E_DATA; v = lens[i - 1]; rep = 3 + b.take(2); }
                else if (sym == 17) rep = 3 + b.take(3);
                else rep = 11 + b.take(7);
                if (i + rep > total) return E_DATA;
                memset(lens + i, v, rep); i += rep;
            }
            if (b.overrun()) return E_INPUT;
            if (lens[256] == 0) return E_DATA;              // no end-of-block code
            if (build_table(lens, hlit, 0, dyn.ll, LL_BITS) || build_table(lens + hlit, hdist, 1, dyn.d, D_BITS)) return E_DATA;
            t = &dyn;
        }
        const uint32_t *const ll = t->ll, *const dt = t->d;
        uint8_t *o = out + pos;
        uint8_t *const o_end = out + cap;
        // ---- fast loop: at least 24 input bytes and 320 output bytes of slack, no bounds checks inside ------------------------
        // Invariant at the top of every iteration: o < o_fast, i.e. more than 320 bytes of output left.  One iteration stores at most
        // 2 + 2 bytes of literals (the second byte of a one-literal entry is scratch) and one match of <= 258 bytes whose 16-byte
        // steps overshoot by <= 15: 277 bytes, all inside the buffer.  The check that re-establishes the invariant is made on the
        // position AFTER the match (o + len), never on the one before it.
        if (b.over == 0) {
            // whole bytes of the careful reader's bit buffer go back to the input: the refill below wants cnt < 64
            const uint8_t *in = b.in - (b.cnt >> 3);
            int cnt = b.cnt & 7;
            uint64_t buf = b.buf & ((1ull << cnt) - 1);
            const uint8_t *const end = b.end;
            uint8_t *const o_fast = cap >= 320 ? o_end - 320 : out;
            bool done = false; int64_t err = 0;
#define REFILL() { buf |= load64(in) << cnt; in += (63 - cnt) >> 3; cnt |= 56; }
#define PUT_LIT(e) { const uint16_t v2 = (uint16_t)((e) >> 16); memcpy(o, &v2, 2); o += ((e) >> 8) & 3; }      /* one or two bytes; the second is scratch when one */
#define LOOKUP_LL(e) { e = ll[buf & ((1u << LL_BITS) - 1)]; if (__builtin_expect(e & K_SUB, 0)) { buf >>= LL_BITS; cnt -= LL_BITS; e = ll[(e >> 16) + (buf & ((1u << ((e >> 8) & 15)) - 1))]; } }
            // Entries of length / distance symbols drop code + extra bits in ONE shift (bits 0-5 hold the sum, see mk_x): the value's
            // extra bits are cut from a copy of the bit buffer off the critical path, which is lookup -> shift -> lookup.
            if (end - in > 24 && o < o_fast) {
                REFILL();
                uint32_t e;
                LOOKUP_LL(e);
                for (;;) {
                    // invariant here: e = entry of the next symbol, looked up in a buffer of >= 56 - 11 bits (second-level hit) else >= 56
                    uint64_t saved = buf;
                    buf >>= (e & 63); cnt -= (e & 63);
                    if (e & K_LIT) {
                        PUT_LIT(e);
                        LOOKUP_LL(e);                                 // >= 41 - 11 bits left: enough for any single code
                        saved = buf;
                        buf >>= (e & 63); cnt -= (e & 63);
                        if (e & K_LIT) {
                            PUT_LIT(e);
                            if (!(end - in > 24 && o < o_fast)) break;
                            REFILL();
                            LOOKUP_LL(e);
                            continue;
                        }
                    }
                    if (__builtin_expect(e & (K_EOB | K_BAD), 0)) { if (e & K_BAD) err = E_DATA; done = true; break; }
                    const uint32_t xl = (e >> 8) & 15;
                    const uint32_t len = (e >> 16) + (uint32_t)((saved >> ((e & 63) - xl)) & ((1u << xl) - 1));
                    if (cnt < 28) REFILL();                           // distance code + extra: <= 15 + 13 bits
                    uint32_t d = dt[buf & ((1u << D_BITS) - 1)];
                    if (__builtin_expect(d & K_SUB, 0)) { buf >>= D_BITS; cnt -= D_BITS; d = dt[(d >> 16) + (buf & ((1u << ((d >> 8) & 15)) - 1))]; }
                    saved = buf;
                    buf >>= (d & 63); cnt -= (d & 63);
                    if (__builtin_expect(d & K_BAD, 0)) { err = E_DATA; done = true; break; }
                    const uint32_t xd = (d >> 8) & 15;
                    const uint32_t dist = (d >> 16) + (uint32_t)((saved >> ((d & 63) - xd)) & ((1u << xd) - 1));
                    if (__builtin_expect(dist > (uint64_t)(o - out), 0)) { err = E_DATA; done = true; break; }
                    // next symbol's entry on its way while the bytes are copied
                    const bool more = end - in > 24 && o + len < o_fast;     // o + len: where the NEXT iteration starts
                    uint32_t e_next = 0;
                    if (more) { REFILL(); LOOKUP_LL(e_next); }
                    const uint8_t *s = o - dist;
                    uint8_t *const oe = o + len;
                    if (dist >= 16 || dist >= len) {                  // no overlap inside one 16-byte step (bytes read at / behind o are scratch)
                        copy16(o, s);
                        if (__builtin_expect(len > 16, 0)) { do { o += 16; s += 16; copy16(o, s); } while (o + 16 < oe); }
                    } else if (dist >= 8) {
                        do { copy8(o, s); o += 8; s += 8; } while (o < oe);
                    } else if (dist == 1 || dist == 2 || dist == 4) { // runs of one byte / int16 / float32 value: the period divides 8
                        uint64_t v;
                        if (dist == 1) v = 0x0101010101010101ull * s[0];
                        else if (dist == 2) { uint16_t t; memcpy(&t, s, 2); v = 0x0001000100010001ull * t; }
                        else { uint32_t t; memcpy(&t, s, 4); v = 0x0000000100000001ull * t; }
                        do { memcpy(o, &v, 8); o += 8; } while (o < oe);
                    } else {
                        do { *o++ = *s++; } while (o < oe);
                    }
                    o = oe;
                    if (!more) break;
                    e = e_next;
                }
            }
#undef LOOKUP_LL
#undef REFILL
#undef PUT_LIT
            // give the whole unread bytes back so that the careful loop starts from a consistent state
            in -= cnt >> 3; cnt &= 7; buf &= (1ull << cnt) - 1;
            b.in = in; b.buf = buf; b.cnt = cnt;
            pos = (uint64_t)(o - out);
            if (err) return err;
            if (done) continue;                             // end of block
        }
        // ---- careful loop: near the end of the input or of the output -------------------------------------------------------
        for (;;) {
            b.fill_safe();
            uint32_t e = ll[b.buf & ((1u << LL_BITS) - 1)];
            if (e & K_SUB) { b.take(LL_BITS); e = ll[(e >> 16) + (b.buf & ((1u << ((e >> 8) & 15)) - 1))]; }
            if (e & K_LIT) {
                b.take(e & 63);
                if (b.overrun()) return E_INPUT;
                const int k = (e >> 8) & 3;
                if (o_end - o < k) return E_OUTPUT;
                *o++ = (uint8_t)(e >> 16);
                if (k == 2) *o++ = (uint8_t)(e >> 24);
                continue;
            }
            if (e & K_BAD) return E_DATA;
            const uint32_t xl = (e >> 8) & 15;
            b.take((e & 63) - xl);
            if (b.overrun()) return E_INPUT;
            if (e & K_EOB) break;
            uint32_t len = (e >> 16) + b.take(xl);
            b.fill_safe();
            uint32_t d = dt[b.buf & ((1u << D_BITS) - 1)];
            if (d & K_SUB) { b.take(D_BITS); d = dt[(d >> 16) + (b.buf & ((1u << ((d >> 8) & 15)) - 1))]; }
            if (d & K_BAD) return E_DATA;
            const uint32_t xd = (d >> 8) & 15;
            b.take((d & 63) - xd);
            const uint32_t dist = (d >> 16) + b.take(xd);
            if (b.overrun()) return E_INPUT;
            if (dist > (uint64_t)(o - out)) return E_DATA;
            if ((uint64_t)(o_end - o) < len) return E_OUTPUT;
            const uint8_t *s = o - dist;
            while (len--) *o++ = *s++;
        }
        pos = (uint64_t)(o - out);
    } while (!last);
    return (int64_t)pos;
}

// ---- CRC-32 (IEEE 802.3, reflected, as gzip uses it) -----------------------------------------------------------------------------
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
};
const CrcTables &crc_tables() { static const CrcTables c; return c; }

uint32_t crc32_slice8(uint32_t crc, const uint8_t *p, size_t n) {          // crc = running value with the usual pre / post inversion outside
    const CrcTables &T = crc_tables();
    while (n && ((uintptr_t)p & 7)) { crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xff]; --n; }
    while (n >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        v ^= crc;
        crc = T.t[7][v & 0xff] ^ T.t[6][(v >> 8) & 0xff] ^ T.t[5][(v >> 16) & 0xff] ^ T.t[4][(v >> 24) & 0xff] ^
              T.t[3][(v >> 32) & 0xff] ^ T.t[2][(v >> 40) & 0xff] ^ T.t[1][(v >> 48) & 0xff] ^ T.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ T.t[0][(crc ^ *p++) & 0xff];
    return crc;
}

#if defined(__x86_64__)
// x^n mod P for the reflected CRC-32 polynomial, in the bit order the carry-less folding below needs: computed, not quoted --
// fold constants are k(n) = (x^n mod P) bit-reflected over 33 bits ... here over 32 bits and shifted left by one.
uint64_t xn_mod_p_reflected(int n) {
    // work in the non-reflected domain: r = x^n mod P(x), P = 0x104C11DB7
    uint32_t r = 1;                                          // x^0
    for (int i = 0; i < n; ++i) r = (r << 1) ^ ((r & 0x80000000u) ? 0x04C11DB7u : 0u);
    // reflect the 32 coefficient bits and shift left by one (the "<< 1" accounts for the 33-bit reflected product alignment)
    uint64_t v = 0;
    for (int i = 0; i < 32; ++i) if (r & (1u << i)) v |= 1ull << (31 - i);
    return v << 1;
}

__attribute__((target("pclmul,sse4.1")))
uint32_t crc32_clmul(uint32_t crc, const uint8_t *p, size_t n) {
    // Folding by four 128-bit lanes (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ"), reflected form.
    static const uint64_t k1 = xn_mod_p_reflected(4 * 128 + 32), k2 = xn_mod_p_reflected(4 * 128 - 32);
    static const uint64_t k3 = xn_mod_p_reflected(128 + 32), k4 = xn_mod_p_reflected(128 - 32);
    if (n < 64 + 16) return crc32_slice8(crc, p, n);
    __m128i x0 = _mm_loadu_si128((const __m128i *)p), x1 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x2 = _mm_loadu_si128((const __m128i *)(p + 32)), x3 = _mm_loadu_si128((const __m128i *)(p + 48));
    x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)crc));
    p += 64; n -= 64;
    const __m128i k12 = _mm_set_epi64x((long long)k2, (long long)k1);
    while (n >= 64) {
        __m128i a0 = _mm_clmulepi64_si128(x0, k12, 0x00), b0 = _mm_clmulepi64_si128(x0, k12, 0x11);
        __m128i a1 = _mm_clmulepi64_si128(x1, k12, 0x00), b1 = _mm_clmulepi64_si128(x1, k12, 0x11);
        __m128i a2 = _mm_clmulepi64_si128(x2, k12, 0x00), b2 = _mm_clmulepi64_si128(x2, k12, 0x11);
        __m128i a3 = _mm_clmulepi64_si128(x3, k12, 0x00), b3 = _mm_clmulepi64_si128(x3, k12, 0x11);
        x0 = _mm_xor_si128(_mm_xor_si128(a0, b0), _mm_loadu_si128((const __m128i *)p));
        x1 = _mm_xor_si128(_mm_xor_si128(a1, b1), _mm_loadu_si128((const __m128i *)(p + 16)));
        x2 = _mm_xor_si128(_mm_xor_si128(a2, b2), _mm_loadu_si128((const __m128i *)(p + 32)));
        x3 = _mm_xor_si128(_mm_xor_si128(a3, b3), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    const __m128i k34 = _mm_set_epi64x((long long)k4, (long long)k3);
#define FOLD1(x, y) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k34, 0x00), _mm_clmulepi64_si128(x, k34, 0x11)), y)
    __m128i x = FOLD1(x0, x1); x = FOLD1(x, x2); x = FOLD1(x, x3);
    while (n >= 16) { x = FOLD1(x, _mm_loadu_si128((const __m128i *)p)); p += 16; n -= 16; }
#undef FOLD1
    // the last 128 bits: finish with the table method over the 16 bytes (plus the tail), starting from a zero CRC state --
    // CRC(state 0, 16 bytes) of the folded remainder equals the CRC state after the original message up to here.
    uint8_t tmp[16];
    _mm_storeu_si128((__m128i *)tmp, x);
    uint32_t c = crc32_slice8(0, tmp, 16);
    return crc32_slice8(c, p, n);
}
#endif

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
#if defined(__x86_64__)
    static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (clmul) return crc32_clmul(crc, p, n);
#endif
    return crc32_slice8(crc, p, n);
}

}  // namespace

extern "C" {

uint32_t ukbb_fcn_gzip_crc(uint32_t crc, const uint8_t *data, uint64_t n) {
    return ~crc32_update(~crc, data, (size_t)n);
}

int64_t ukbb_fcn_gunzip(const uint8_t *src, uint64_t src_len, uint8_t *dst, uint64_t dst_cap, int verify_crc) {
    if (!src || (!dst && dst_cap)) return UKBB_EINVAL;
    const uint8_t *p = src, *const end = src + src_len;
    uint64_t pos = 0;
    int members = 0;
    for (;;) {
        // ---- member header (RFC 1952 2.3) ----
        if (end - p < 18) return members ? UKBB_EINVAL : UKBB_EINVAL;
        if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8) return UKBB_EINVAL;
        const uint8_t flg = p[3];
        if (flg & 0xe2) return UKBB_EINVAL;                 // reserved bits, or FHCRC (header CRC: left to zlib)
        p += 10;
        if (flg & 4) {                                      // FEXTRA
            if (end - p < 2) return UKBB_EINVAL;
            const size_t xlen = p[0] | (p[1] << 8);
            p += 2;
            if ((size_t)(end - p) < xlen) return UKBB_EINVAL;
            p += xlen;
        }
        for (int f = 8; f <= 16; f <<= 1)                   // FNAME, FCOMMENT: zero-terminated
            if (flg & f) {
                const void *z = memchr(p, 0, (size_t)(end - p));
                if (!z) return UKBB_EINVAL;
                p = (const uint8_t *)z + 1;
            }
        Bits b; b.in = p; b.end = end;
        const uint64_t start = pos;
        const int64_t r = inflate_stream(b, dst, pos, dst_cap);
        if (r < 0) return r == E_OUTPUT ? UKBB_ENOMEM : UKBB_EINVAL;
        pos = (uint64_t)r;
        // trailer: whole bytes left in the bit buffer go back first
        b.take(b.cnt & 7);
        int back = (b.cnt >> 3) - b.over;
        p = b.in - (back > 0 ? back : 0);
        if (end - p < 8) return UKBB_EINVAL;
        const uint32_t want_crc = p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24);
        const uint32_t want_len = p[4] | (p[5] << 8) | (p[6] << 16) | ((uint32_t)p[7] << 24);
        p += 8;
        if ((uint32_t)(pos - start) != want_len) return UKBB_EINVAL;
        if (verify_crc && ukbb_fcn_gzip_crc(0, dst + start, pos - start) != want_crc) return UKBB_EINVAL;
        ++members;
        while (p < end && *p == 0) ++p;                     // zero padding between / after members (tape blocks, dd)
        if (p == end) return (int64_t)pos;
    }
}

}  // extern "C"
