// gzip writer for label volumes (host only, no HIP).
//
// The sequence loop of the reference ends with nib.save of np.zeros(image.shape) filled with the predicted labels
// (common/deploy_network.py:92,116,136-138): 160 MB of float64 per short-axis subject whose voxels take one of <= 4 values,
// pushed through zlib.  With the network at 11 ms per subject that deflate (0.5 s of a core at nibabel's level 1) is what a
// deployment waits for.  This encoder writes the SAME uncompressed stream -- header bytes followed by the labels converted
// to the file's voxel type -- as one gzip member without ever forming it: a run of equal labels is a run of equal E-byte
// patterns, i.e. E literals followed by deflate matches of distance E (RFC 1951, fixed Huffman codes, 14 bits per 258
// bytes for float64), and the CRC-32 of a run of zero bytes is a multiplication by x^(8n) in GF(2)[x]/P.  Any inflater
// (nibabel, zlib's gzread, MIRTK) reads the result as the file the reference writes.
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

}  // namespace

extern "C" {

uint64_t ukbb_fcn_gzip_labels_bound(uint64_t n_voxels, int nifti_datatype, uint64_t prefix_len) {
    uint8_t tmp[8];
    const int e = element_pattern(nifti_datatype, 0, tmp);
    const uint64_t raw = prefix_len + n_voxels * (uint64_t)(e ? e : 8);
    return raw + raw / 8 + 64;                         // <= 9 bits per literal byte, header, trailer, bit-buffer slack
}

int64_t ukbb_fcn_gzip_labels(const uint8_t *labels, uint64_t n_voxels, int nifti_datatype, const uint8_t *prefix, uint64_t prefix_len,
                             uint8_t *out, uint64_t out_cap) {
    if ((!labels && n_voxels) || (!prefix && prefix_len) || !out) return UKBB_EINVAL;
    uint8_t pat[256][8];
    bool have[256] = {false};
    struct Chunk { uint32_t bits; int len; };
    Chunk first_tok[256][3];                           // the E literals that open a run of label v, packed into <= 32-bit pieces
    int first_n[256];
    uint8_t probe[8];
    const int E = element_pattern(nifti_datatype, 0, probe);
    if (!E) return UKBB_EINVAL;
    if (out_cap < 32) return UKBB_ENOMEM;
    const Tables &t = tables();
    // gzip header as Python's GzipFile(filename='', mtime=0, compresslevel=1) writes it (ukbb_cardiac_amd/nifti.py)
    static const uint8_t head[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 4, 0xff};
    memcpy(out, head, 10);
    BitWriter w{out + 10, out + out_cap - 8};
    w.put(1, 1);                                       // BFINAL
    w.put(1, 2);                                       // BTYPE = 01, fixed Huffman
    auto literal = [&](uint8_t b) { w.put(t.lit_bits[b], t.lit_len[b]); };
    // distance code of E: 1 -> 0, 2 -> 1, 4 -> 3, 8 -> code 5 + extra bit 1 (distances 7-8); 5-bit codes go out MSB first
    uint32_t dbits; int dlen;
    {
        const int dcode = E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 3 : 5;
        uint32_t r = 0;
        for (int i = 0; i < 5; ++i) r |= ((dcode >> i) & 1u) << (4 - i);
        dbits = r; dlen = 5;
        if (E == 8) { dbits |= 1u << 5; dlen = 6; }
    }
    auto match = [&](int len) {                        // (length, distance E)
        const int s = t.len_sym[len];
        w.put(t.lit_bits[s], t.lit_len[s]);
        if (t.len_xbits[len]) w.put(t.len_xval[len], t.len_xbits[len]);
        w.put(dbits, dlen);
    };
    const uint32_t tok258 = (uint32_t)t.lit_bits[285] | (dbits << t.lit_len[285]);
    const int tok258_len = t.lit_len[285] + dlen;
    uint32_t reg = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < prefix_len; ++i) literal(prefix[i]);
    reg = crc_bytes(t, reg, prefix, (size_t)prefix_len);
    uint64_t i = 0;
    while (i < n_voxels) {
        const uint8_t v = labels[i];
        uint64_t j = i + 1;
        {                                              // end of the run: bytes, then 8 at a time, then bytes
            const uint64_t splat = 0x0101010101010101ull * v;
            while (j < n_voxels && (j & 7) && labels[j] == v) ++j;
            if (j < n_voxels && !(j & 7)) {
                uint64_t wv;
                while (j + 8 <= n_voxels && (memcpy(&wv, labels + j, 8), wv == splat)) j += 8;
                while (j < n_voxels && labels[j] == v) ++j;
            }
        }
        if (!have[v]) {
            element_pattern(nifti_datatype, v, pat[v]);
            have[v] = true;
            int nchunk = 0;
            Chunk c{0, 0};
            for (int b = 0; b < E; ++b) {
                const uint32_t lb = t.lit_bits[pat[v][b]];
                const int ll = t.lit_len[pat[v][b]];
                if (c.len + ll > 32) { first_tok[v][nchunk++] = c; c = Chunk{0, 0}; }
                c.bits |= lb << c.len; c.len += ll;
            }
            first_tok[v][nchunk++] = c;
            first_n[v] = nchunk;
        }
        const uint8_t *P = pat[v];
        const uint64_t run = j - i;
        // ---- deflate tokens ----
        for (int c = 0; c < first_n[v]; ++c) w.put(first_tok[v][c].bits, first_tok[v][c].len);
        uint64_t R = (run - 1) * (uint64_t)E;
        while (R >= 258 + 3 || R == 258) { w.put(tok258, tok258_len); R -= 258; }
        if (R > 258) { match((int)(R - 3)); R = 3; }   // 259, 260: leave a legal match of 3
        if (R >= 3) match((int)R);
        else for (uint64_t b = 0; b < R; ++b) literal(P[b % E]);
        if (w.overflow) return UKBB_ENOMEM;
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
        i = j;
    }
    w.put(t.lit_bits[256], t.lit_len[256]);            // end of block
    w.finish();
    if (w.overflow) return UKBB_ENOMEM;
    uint8_t *p = w.p;
    const uint32_t crc = reg ^ 0xFFFFFFFFu;
    const uint32_t isize = (uint32_t)((prefix_len + n_voxels * (uint64_t)E) & 0xFFFFFFFFull);
    memcpy(p, &crc, 4); memcpy(p + 4, &isize, 4);
    return (int64_t)(p + 8 - out);
}

}  // extern "C"
