// gzip writer for label volumes (host only, no HIP).
//
// The sequence loop of the reference ends with nib.save of np.zeros(image.shape) filled with the predicted labels
// (common/deploy_network.py:92,116,136-138): 160 MB of float64 per short-axis subject whose voxels take one of <= 4 values,
// pushed through zlib.  With the network at 11 ms per subject that deflate (0.5 s of a core at nibabel's level 1) is what a
// deployment waits for.  This encoder writes the SAME uncompressed stream -- header bytes followed by the labels converted
// to the file's voxel type -- as one gzip member without ever forming it: a run of equal labels is a run of equal E-byte
// patterns, i.e. E literals followed by deflate matches of distance E (RFC 1951), and the CRC-32 of a run of zero bytes is a
// multiplication by x^(8n) in GF(2)[x]/P.  Any inflater (nibabel, zlib's gzread, MIRTK) reads the result as the file the
// reference writes.
//
// Two code sets for the same token stream: UKBB_GZIP_FIXED (BTYPE 01: 14 bits per 258 bytes of float64, ~80 bits per
// short run) and UKBB_GZIP_DYNAMIC (BTYPE 10, the default): one counting pass over the runs gives the exact token
// histogram, from which a length-limited Huffman code is built -- the handful of byte values a label pattern consists of
// and the one distance in use then cost 1-3 bits each, which brings the file to the size zlib level 1 reaches on the
// same volume (r03: smaller on every volume tried) for one extra 20 MB read.
#include "../../include/ukbb_fcn.h"

#include <cstdint>
#include <cstring>

namespace {

constexpr uint32_t POLY = 0xEDB88320u;                 // CRC-32 (reflected)

struct Tables {
    uint32_t crc[8][256];                              // slicing-by-8
    uint32_t x2n[32];                                  // x^(2^k) mod P, reflected
    uint16_t lit_bits[288]; uint8_t lit_len[288];     // fixed literal/length codes, bit-reversed for LSB-first output
    uint16_t len_sym[259]; uint8_t len_xbits[259]; uint16_t len_xval[259];
    Tables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = c & 1 ? (c >> 1) ^ POLY : c >> 1;
            crc[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) crc[t][i] = (crc[t - 1][i] >> 8) ^ crc[0][crc[t - 1][i] & 0xff];
        x2n[0] = 1u << 30;                             // x^1
        for (int k = 1; k < 32; ++k) x2n[k] = mult(x2n[k - 1], x2n[k - 1]);
        auto rev = [](uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; } return r; };
        for (int s = 0; s < 288; ++s) {
            uint32_t code; int len;
            if (s < 144) { code = 0x30 + s; len = 8; }
            else if (s < 256) { code = 0x190 + (s - 144); len = 9; }
            else if (s < 280) { code = s - 256; len = 7; }
            else { code = 0xC0 + (s - 280); len = 8; }
            lit_bits[s] = (uint16_t)rev(code, len); lit_len[s] = (uint8_t)len;
        }
        static const int base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const int xb[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        for (int l = 3; l <= 258; ++l) {
            int i = 28;
            while (base[i] > l) --i;
            if (l == 258) i = 28;
            else if (i == 28) i = 27;                  // 227..257 belong to symbol 284
            len_sym[l] = (uint16_t)(257 + i); len_xbits[l] = (uint8_t)xb[i]; len_xval[l] = (uint16_t)(l - base[i]);
        }
    }
    static uint32_t mult(uint32_t a, uint32_t b) {     // a * b mod P (reflected polynomials, zlib's multmodp)
        uint32_t m = 1u << 31, p = 0;
        for (;;) {
            if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
            m >>= 1;
            b = b & 1 ? (b >> 1) ^ POLY : b >> 1;
        }
        return p;
    }
    uint32_t shift_zero_bytes(uint32_t reg, uint64_t nbytes) const {   // CRC register after nbytes zero bytes: reg * x^(8 n)
        uint32_t p = 1u << 31;                         // x^0
        unsigned k = 3;
        for (uint64_t n = nbytes; n; n >>= 1, ++k)
            if (n & 1) p = mult(x2n[k & 31], p);
        return mult(p, reg);
    }
};
const Tables &tables() { static const Tables t; return t; }

struct BitWriter {
    uint8_t *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    inline void put(uint32_t v, int bits) {
        acc |= (uint64_t)v << n;
        n += bits;
        if (n >= 32) {
            if (end - p >= 4) { memcpy(p, &acc, 4); p += 4; } else { overflow = true; }
            acc >>= 32; n -= 32;
        }
    }
    void finish() {
        while (n > 0) { if (p < end) *p++ = (uint8_t)acc; else overflow = true; acc >>= 8; n -= 8; }
        n = 0;
    }
};

inline uint32_t crc_bytes(const Tables &t, uint32_t reg, const uint8_t *b, size_t n) {
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, b, 4); memcpy(&hi, b + 4, 4);
        lo ^= reg;
        reg = t.crc[7][lo & 0xff] ^ t.crc[6][(lo >> 8) & 0xff] ^ t.crc[5][(lo >> 16) & 0xff] ^ t.crc[4][lo >> 24] ^
              t.crc[3][hi & 0xff] ^ t.crc[2][(hi >> 8) & 0xff] ^ t.crc[1][(hi >> 16) & 0xff] ^ t.crc[0][hi >> 24];
        b += 8; n -= 8;
    }
    while (n--) reg = (reg >> 8) ^ t.crc[0][(reg ^ *b++) & 0xff];
    return reg;
}

int element_pattern(int datatype, unsigned label, uint8_t out[8]) {   // little-endian bytes of `label` as NIfTI datatype; returns the size
    switch (datatype) {
        case 2:  out[0] = (uint8_t)label; return 1;                                         // uint8
        case 4:  { const int16_t v = (int16_t)label; memcpy(out, &v, 2); return 2; }         // int16
        case 8:  { const int32_t v = (int32_t)label; memcpy(out, &v, 4); return 4; }         // int32
        case 16: { const float v = (float)label; memcpy(out, &v, 4); return 4; }             // float32
        case 64: { const double v = (double)label; memcpy(out, &v, 8); return 8; }           // float64
        default: return 0;
    }
}


// ---- code sets ---------------------------------------------------------------------------------------------------------
struct CodeSet {
    uint16_t lit_bits[288]; uint8_t lit_len[288];     // literal/length codes, bit-reversed for LSB-first output
    uint32_t dbits; int dlen;                          // the one distance in use (E), extra bit included
};

inline uint32_t rev_bits(uint32_t v, int n) { uint32_t r = 0; for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; } return r; }

// Huffman code lengths for freq[0..n) limited to max_len bits; symbols with freq 0 get length 0.  At least two symbols
// get a code (zlib's inflate rejects an incomplete literal/length or code-length code).
void huffman_lengths(const uint64_t *freq, int n, int max_len, uint8_t *len) {
    int sym[288 + 2], m = 0;
    for (int i = 0; i < n; ++i) { len[i] = 0; if (freq[i]) sym[m++] = i; }
    for (int i = 0; m < 2 && i < n; ++i) if (!freq[i]) sym[m++] = i;    // pad with unused symbols (weight 0 -> treated as 1)
    // plain O(m^2) Huffman over <= 288 leaves: parent links, depth = code length
    uint64_t w[2 * 290]; int parent[2 * 290]; bool alive[2 * 290];
    int nn = m;
    for (int i = 0; i < m; ++i) { w[i] = freq[sym[i]] ? freq[sym[i]] : 1; parent[i] = -1; alive[i] = true; }
    for (int left = m; left > 1; --left) {
        int a = -1, b = -1;
        for (int i = 0; i < nn; ++i) if (alive[i]) {
            if (a < 0 || w[i] < w[a]) { b = a; a = i; }
            else if (b < 0 || w[i] < w[b]) b = i;
        }
        w[nn] = w[a] + w[b]; parent[nn] = -1; alive[nn] = true;
        parent[a] = parent[b] = nn; alive[a] = alive[b] = false;
        ++nn;
    }
    int L[290];
    for (int i = 0; i < m; ++i) { int d = 0; for (int j = i; parent[j] >= 0; j = parent[j]) ++d; L[i] = d < 1 ? 1 : d; }
    // limit: clamp, then repair the Kraft sum (unit 2^-max_len) to exactly 1
    int64_t K = 0;
    const int64_t full = (int64_t)1 << max_len;
    for (int i = 0; i < m; ++i) { if (L[i] > max_len) L[i] = max_len; K += full >> L[i]; }
    while (K > full) {                                 // lengthen the cheapest symbol that still can be (longest code < max_len, smallest weight)
        int best = -1;
        for (int i = 0; i < m; ++i) if (L[i] < max_len && (best < 0 || L[i] > L[best] || (L[i] == L[best] && w[i] < w[best]))) best = i;
        K -= full >> (L[best] + 1); ++L[best];
    }
    while (K < full) {                                 // shorten: the heaviest symbol whose step fits the deficit
        int best = -1;
        for (int i = 0; i < m; ++i) if (L[i] > 1 && (full >> L[i]) <= full - K && (best < 0 || w[i] > w[best])) best = i;
        K += full >> L[best]; --L[best];
    }
    for (int i = 0; i < m; ++i) len[sym[i]] = (uint8_t)L[i];
}

void canonical_codes(const uint8_t *len, int n, int max_len, uint16_t *bits) {   // RFC 1951 3.2.2, bit-reversed
    int count[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) ++count[len[i]];
    count[0] = 0;
    int code = 0;
    for (int b = 1; b <= max_len; ++b) { code = (code + count[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < n; ++i) bits[i] = len[i] ? (uint16_t)rev_bits((uint32_t)next[len[i]]++, len[i]) : 0;
}

int dist_code_of(int E) { return E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 3 : 5; }

void fixed_codes(const Tables &t, int E, CodeSet &c) {
    for (int s = 0; s < 288; ++s) { c.lit_bits[s] = t.lit_bits[s]; c.lit_len[s] = t.lit_len[s]; }
    // distance code of E: 1 -> 0, 2 -> 1, 4 -> 3, 8 -> code 5 + extra bit 1 (distances 7-8); 5-bit codes go out MSB first
    c.dbits = rev_bits((uint32_t)dist_code_of(E), 5); c.dlen = 5;
    if (E == 8) { c.dbits |= 1u << 5; c.dlen = 6; }
}

// Walks the runs of equal labels: f(v, run_length) per run, in order.
template <class F>
inline void for_each_run(const uint8_t *labels, uint64_t n_voxels, F &&f) {
    uint64_t i = 0;
    while (i < n_voxels) {
        const uint8_t v = labels[i];
        uint64_t j = i + 1;
        const uint64_t splat = 0x0101010101010101ull * v;   // end of the run: bytes, then 8 at a time, then bytes
        while (j < n_voxels && (j & 7) && labels[j] == v) ++j;
        if (j < n_voxels && !(j & 7)) {
            uint64_t wv;
            while (j + 8 <= n_voxels && (memcpy(&wv, labels + j, 8), wv == splat)) j += 8;
            while (j < n_voxels && labels[j] == v) ++j;
        }
        f(v, j - i);
        i = j;
    }
}

// The matches that follow the E opening literals of a run: R = (run - 1) * E bytes at distance E as n258 matches of
// length 258 plus a tail: 0, a literal tail of 1-2 bytes (tail < 3), one match (3..258) or two (259, 260 -> tail - 3, 3).
inline void split_run(uint64_t R, uint64_t &n258, unsigned &tail) {
    n258 = 0;
    if (R >= 261) { n258 = (R - 3) / 258; R -= 258 * n258; }
    if (R == 258) { ++n258; R = 0; }
    tail = (unsigned)R;
}

}  // namespace

extern "C" {

uint64_t ukbb_fcn_gzip_labels_bound(uint64_t n_voxels, int nifti_datatype, uint64_t prefix_len) {
    uint8_t tmp[8];
    const int e = element_pattern(nifti_datatype, 0, tmp);
    const uint64_t raw = prefix_len + n_voxels * (uint64_t)(e ? e : 8);
    return raw * 2 + 1024;                             // <= 15 bits per literal byte, block header, trailer, bit-buffer slack
}

int64_t ukbb_fcn_gzip_labels_mode(const uint8_t *labels, uint64_t n_voxels, int nifti_datatype, const uint8_t *prefix, uint64_t prefix_len,
                                  uint8_t *out, uint64_t out_cap, int mode) {
    if ((!labels && n_voxels) || (!prefix && prefix_len) || !out) return UKBB_EINVAL;
    if (mode != UKBB_GZIP_FIXED && mode != UKBB_GZIP_DYNAMIC) return UKBB_EINVAL;
    uint8_t pat[256][8];
    uint8_t probe[8];
    const int E = element_pattern(nifti_datatype, 0, probe);
    if (!E) return UKBB_EINVAL;
    if (out_cap < 512) return UKBB_ENOMEM;
    for (unsigned v = 0; v < 256; ++v) element_pattern(nifti_datatype, v, pat[v]);
    const Tables &t = tables();
    CodeSet cs;
    // gzip header as Python's GzipFile(filename='', mtime=0, compresslevel=1) writes it (ukbb_cardiac_amd/nifti.py)
    static const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 0xff};
    memcpy(out, head, 10);
    BitWriter w{out + 10, out + out_cap - 8};
    w.put(1, 1);                                       // BFINAL
    if (mode == UKBB_GZIP_FIXED) {
        w.put(1, 2);                                   // BTYPE = 01, fixed Huffman
        fixed_codes(t, E, cs);
    } else {
        // ---- pass 1: the exact histogram of the tokens pass 2 will write ----
        uint64_t runs[256] = {0}, lit[288] = {0};
        for (uint64_t i = 0; i < prefix_len; ++i) ++lit[prefix[i]];
        for_each_run(labels, n_voxels, [&](uint8_t v, uint64_t run) {
            ++runs[v];
            uint64_t n258; unsigned tail;
            split_run((run - 1) * (uint64_t)E, n258, tail);
            lit[285] += n258;
            if (tail > 258) { ++lit[t.len_sym[tail - 3]]; ++lit[t.len_sym[3]]; }
            else if (tail >= 3) ++lit[t.len_sym[tail]];
            else for (unsigned b = 0; b < tail; ++b) ++lit[pat[v][b % E]];
        });
        for (unsigned v = 0; v < 256; ++v)
            if (runs[v]) for (int b = 0; b < E; ++b) lit[pat[v][b]] += runs[v];
        lit[256] = 1;                                  // end of block
        huffman_lengths(lit, 286, 15, cs.lit_len);
        cs.lit_len[286] = cs.lit_len[287] = 0;
        canonical_codes(cs.lit_len, 286, 15, cs.lit_bits);
        // one distance code in use: a single 1-bit code (RFC 1951 3.2.7), then the extra bit of distances 7-8 for E = 8
        const int dcode = dist_code_of(E);
        cs.dbits = 0; cs.dlen = 1;
        if (E == 8) { cs.dbits |= 1u << 1; cs.dlen = 2; }
        // ---- block header: BTYPE 10, HLIT / HDIST / HCLEN, code-length code, run-length coded code lengths ----
        int nlit = 286;
        while (nlit > 257 && !cs.lit_len[nlit - 1]) --nlit;
        const int ndist = dcode + 1;
        uint8_t seq[286 + 30];
        for (int i = 0; i < nlit; ++i) seq[i] = cs.lit_len[i];
        for (int i = 0; i < ndist; ++i) seq[nlit + i] = (uint8_t)(i == dcode ? 1 : 0);
        const int nseq = nlit + ndist;
        struct Cl { uint8_t sym, xbits, xval; } cl[286 + 30];
        int ncl = 0;
        uint64_t clfreq[19] = {0};
        for (int i = 0; i < nseq;) {
            int j = i + 1;
            while (j < nseq && seq[j] == seq[i]) ++j;
            int rep = j - i;
            if (seq[i] == 0) {
                while (rep >= 11) { const int k = rep > 138 ? 138 : rep; cl[ncl++] = Cl{18, 7, (uint8_t)(k - 11)}; rep -= k; }
                if (rep >= 3) { cl[ncl++] = Cl{17, 3, (uint8_t)(rep - 3)}; rep = 0; }
                while (rep--) cl[ncl++] = Cl{0, 0, 0};
            } else {
                cl[ncl++] = Cl{seq[i], 0, 0}; --rep;
                while (rep >= 3) { const int k = rep > 6 ? 6 : rep; cl[ncl++] = Cl{16, 2, (uint8_t)(k - 3)}; rep -= k; }
                while (rep--) cl[ncl++] = Cl{seq[i], 0, 0};
            }
            i = j;
        }
        for (int i = 0; i < ncl; ++i) ++clfreq[cl[i].sym];
        uint8_t cll[19]; uint16_t clb[19];
        huffman_lengths(clfreq, 19, 7, cll);
        canonical_codes(cll, 19, 7, clb);
        static const int order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19;
        while (hclen > 4 && !cll[order[hclen - 1]]) --hclen;
        w.put(2, 2);                                   // BTYPE = 10, dynamic Huffman
        w.put((uint32_t)(nlit - 257), 5);
        w.put((uint32_t)(ndist - 1), 5);
        w.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; ++i) w.put(cll[order[i]], 3);
        for (int i = 0; i < ncl; ++i) {
            w.put(clb[cl[i].sym], cll[cl[i].sym]);
            if (cl[i].xbits) w.put(cl[i].xval, cl[i].xbits);
        }
    }
    auto literal = [&](uint8_t b) { w.put(cs.lit_bits[b], cs.lit_len[b]); };
    auto match = [&](int len) {                        // (length, distance E)
        const int s = t.len_sym[len];
        w.put(cs.lit_bits[s], cs.lit_len[s]);
        if (t.len_xbits[len]) w.put(t.len_xval[len], t.len_xbits[len]);
        w.put(cs.dbits, cs.dlen);
    };
    // (258, E) tokens, several to a 32-bit piece
    const int tok258_len = cs.lit_len[285] + cs.dlen;
    const uint32_t tok258 = (uint32_t)cs.lit_bits[285] | (cs.dbits << cs.lit_len[285]);
    int pack_n = 32 / (tok258_len ? tok258_len : 1);
    if (pack_n < 1) pack_n = 1;
    uint32_t pack258 = 0;
    for (int k = 0; k < pack_n && pack_n * tok258_len <= 32; ++k) pack258 |= tok258 << (k * tok258_len);
    struct Chunk { uint32_t bits; int len; };
    Chunk first_tok[256][5];                           // the E literals that open a run of label v, packed into <= 32-bit pieces
    int first_n[256];
    bool have[256] = {false};
    uint32_t reg = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < prefix_len; ++i) literal(prefix[i]);
    reg = crc_bytes(t, reg, prefix, (size_t)prefix_len);
    bool overflow = false;
    for_each_run(labels, n_voxels, [&](uint8_t v, uint64_t run) {
        if (overflow) return;
        if (!have[v]) {
            have[v] = true;
            int nchunk = 0;
            Chunk c{0, 0};
            for (int b = 0; b < E; ++b) {
                const uint32_t lb = cs.lit_bits[pat[v][b]];
                const int ll = cs.lit_len[pat[v][b]];
                if (c.len + ll > 32) { first_tok[v][nchunk++] = c; c = Chunk{0, 0}; }
                c.bits |= lb << c.len; c.len += ll;
            }
            first_tok[v][nchunk++] = c;
            first_n[v] = nchunk;
        }
        const uint8_t *P = pat[v];
        // ---- deflate tokens ----
        for (int c = 0; c < first_n[v]; ++c) w.put(first_tok[v][c].bits, first_tok[v][c].len);
        uint64_t n258; unsigned tail;
        split_run((run - 1) * (uint64_t)E, n258, tail);
        for (; n258 >= (uint64_t)pack_n; n258 -= pack_n) w.put(pack258, pack_n * tok258_len);
        for (; n258; --n258) w.put(tok258, tok258_len);
        if (tail > 258) { match((int)tail - 3); match(3); }
        else if (tail >= 3) match((int)tail);
        else for (unsigned b = 0; b < tail; ++b) literal(P[b % E]);
        if (w.overflow) { overflow = true; return; }
        // ---- CRC-32 of the run ----
        bool zero = true;
        for (int b = 0; b < E; ++b) zero = zero && P[b] == 0;
        const uint64_t nbytes = run * (uint64_t)E;
        if (zero && nbytes >= 512) reg = t.shift_zero_bytes(reg, nbytes);
        else if (E == 8) {
            uint32_t lo0, hi;
            memcpy(&lo0, P, 4); memcpy(&hi, P + 4, 4);
            const uint32_t h = t.crc[3][hi & 0xff] ^ t.crc[2][(hi >> 8) & 0xff] ^ t.crc[1][(hi >> 16) & 0xff] ^ t.crc[0][hi >> 24];
            for (uint64_t k = 0; k < run; ++k) {
                const uint32_t lo = lo0 ^ reg;
                reg = t.crc[7][lo & 0xff] ^ t.crc[6][(lo >> 8) & 0xff] ^ t.crc[5][(lo >> 16) & 0xff] ^ t.crc[4][lo >> 24] ^ h;
            }
        } else {
            for (uint64_t k = 0; k < run; ++k) reg = crc_bytes(t, reg, P, (size_t)E);
        }
    });
    if (overflow) return UKBB_ENOMEM;
    w.put(cs.lit_bits[256], cs.lit_len[256]);          // end of block
    w.finish();
    if (w.overflow) return UKBB_ENOMEM;
    uint8_t *p = w.p;
    const uint32_t crc = reg ^ 0xFFFFFFFFu;
    const uint32_t isize = (uint32_t)((prefix_len + n_voxels * (uint64_t)E) & 0xFFFFFFFFull);
    memcpy(p, &crc, 4); memcpy(p + 4, &isize, 4);
    return (int64_t)(p + 8 - out);
}

int64_t ukbb_fcn_gzip_labels(const uint8_t *labels, uint64_t n_voxels, int nifti_datatype, const uint8_t *prefix, uint64_t prefix_len,
                             uint8_t *out, uint64_t out_cap) {
    return ukbb_fcn_gzip_labels_mode(labels, n_voxels, nifti_datatype, prefix, prefix_len, out, out_cap, UKBB_GZIP_DYNAMIC);
}

}  // extern "C"
