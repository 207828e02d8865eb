// png_io.hpp -- the image-file side of the callers (cv::imread(..., CV_LOAD_IMAGE_GRAYSCALE) at
// sgm_single/main.cc:20-21, cv::imwrite of CV_8U / CV_16U at :54-57) without OpenCV or libpng:
// a small PNG codec on zlib.  Reads every non-interlaced colour type and bit depth and converts to
// 8-bit gray (16-bit samples keep their high byte; colour uses the fixed-point BGR2GRAY weights
// 4899/9617/1868 >> 14); read16() keeps 16-bit gray samples, the KITTI disparity format
// (reconstruction/evaluation/io_disp.h:220-238).  Writes 8- and 16-bit gray.
#ifndef SGMX_PNG_IO_HPP
#define SGMX_PNG_IO_HPP

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <zlib.h>

#include "sgmx_recon.hpp"

namespace sgmx {
namespace png {

inline uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put32(std::vector<unsigned char> &v, uint32_t x)
{
    v.push_back((unsigned char)(x >> 24));
    v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8));
    v.push_back((unsigned char)x);
}

inline void slurp_into(const std::string &path, std::vector<unsigned char> &buf)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open " + path);
    // sized once from the file length when the stream can tell it, grown by reading otherwise
    long len = -1;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        len = std::ftell(f);
        std::rewind(f);
    }
    buf.resize(len > 0 ? (size_t)len : (size_t)1 << 16);
    size_t have = 0, n;
    while ((n = std::fread(buf.data() + have, 1, buf.size() - have, f)) > 0) {
        have += n;
        if (have == buf.size())
            buf.resize(buf.size() * 2);
    }
    std::fclose(f);
    buf.resize(have);
}

inline std::vector<unsigned char> slurp(const std::string &path)
{
    std::vector<unsigned char> buf;
    slurp_into(path, buf);
    return buf;
}

// ---------------------------------------------------------------------------
// Inflate for the reader: a table-driven decoder for whole zlib streams with a known output size
// (a PNG's (rowbytes + 1) * H).  Literal / length codes up to 11 bits and distance codes up to 8
// bits resolve in one table lookup from a 64-bit bit buffer; the rare longer codes take a canonical
// bit-by-bit walk.  Every read and write is bounds-checked; the Adler-32 trailer is verified.  On
// anything it does not accept it returns false and decode() hands the stream to zlib's uncompress,
// which then gives the verdict.  (1.2-1.7x the speed of zlib 1.2.11's inflate on KITTI-size 8-bit images.)
// Little-endian hosts.
// ---------------------------------------------------------------------------
struct HuffDec {
    uint16_t count[16];  // codes per length
    uint16_t symbol[288]; // symbols in canonical order
    uint16_t fast[2048]; // (symbol << 4) | length for codes of <= fast_bits bits, 0 = longer / unused
    int fast_bits;
};

inline bool huff_build(HuffDec &h, const unsigned char *len, int n, int fast_bits)
{
    h.fast_bits = fast_bits;
    std::memset(h.count, 0, sizeof(h.count));
    std::memset(h.fast, 0, sizeof(uint16_t) << fast_bits);
    for (int i = 0; i < n; i++)
        h.count[len[i]]++;
    h.count[0] = 0;
    int left = 1;
    for (int l = 1; l < 16; l++) {
        left = (left << 1) - h.count[l];
        if (left < 0)
            return false; // over-subscribed
    }
    uint16_t offs[16], next[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++)
        offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    for (int i = 0; i < n; i++)
        if (len[i])
            h.symbol[offs[len[i]]++] = (uint16_t)i;
    unsigned c = 0;
    for (int l = 1; l < 16; l++) { // first canonical code of each length
        next[l] = (uint16_t)c;
        c = (c + h.count[l]) << 1;
    }
    for (int i = 0; i < n; i++) {
        const int l = len[i];
        if (!l || l > fast_bits)
            continue;
        unsigned v = next[l]++, r = 0;
        for (int b = 0; b < l; b++)
            r |= ((v >> b) & 1u) << (l - 1 - b);
        for (unsigned k = r; k < (1u << fast_bits); k += 1u << l)
            h.fast[k] = (uint16_t)((i << 4) | l);
    }
    return true;
}

// Bit-by-bit canonical decode from the low bits of `bits` (`avail` of them valid): symbol, or -1.
inline int huff_slow(const HuffDec &h, uint64_t bits, int avail, int *used)
{
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16 && l <= avail; l++) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int cnt = h.count[l];
        if (code - cnt < first) {
            *used = l;
            return h.symbol[index + (code - first)];
        }
        index += cnt;
        first = (first + cnt) << 1;
        code <<= 1;
    }
    return -1;
}

inline bool inflate_zlib_exact(const unsigned char *in, size_t in_len, unsigned char *out, size_t out_len)
{
    static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    if (in_len < 6 || (in[0] & 0x0f) != 8 || (in[0] >> 4) > 7 || (in[1] & 0x20) || ((in[0] << 8) | in[1]) % 31 != 0)
        return false;
    const unsigned char *ip = in + 2, *const iend = in + in_len - 4; // the trailer is not deflate data
    unsigned char *op = out, *const oend = out + out_len;
    uint64_t bb = 0;
    int bn = 0;
    // After a refill at least 56 bits are valid (fewer only at the end of the input).  The 8-byte form may leave
    // real, not yet counted input bits above bit `bn`; the next refill ORs the same bits there again.
    auto refill = [&]() {
        if (iend - ip >= 8) {
            uint64_t w;
            std::memcpy(&w, ip, 8);
            bb |= w << bn;
            ip += (63 - bn) >> 3;
            bn |= 56;
        } else {
            if (bn < 64)
                bb &= (1ull << bn) - 1;
            while (bn <= 56 && ip < iend) {
                bb |= (uint64_t)*ip++ << bn;
                bn += 8;
            }
        }
    };
    auto take = [&](int n) {
        const uint32_t v = (uint32_t)(bb & ((1ull << n) - 1));
        bb >>= n;
        bn -= n;
        return v;
    };
    HuffDec lit, dist;
    unsigned char lens[320];
    for (bool last = false; !last;) {
        refill();
        if (bn < 3)
            return false;
        last = take(1) != 0;
        const uint32_t type = take(2);
        if (type == 0) { // stored
            take(bn & 7);
            refill();
            if (bn < 32)
                return false;
            const uint32_t n = take(16), nn = take(16);
            if ((n ^ nn) != 0xffffu)
                return false;
            // bytes still in the bit buffer belong to the input again
            ip -= bn >> 3;
            bb = 0;
            bn = 0;
            if ((size_t)(iend - ip) < n || (size_t)(oend - op) < n)
                return false;
            std::memcpy(op, ip, n);
            op += n;
            ip += n;
            continue;
        }
        if (type == 3)
            return false;
        if (type == 1) { // fixed codes
            for (int i = 0; i < 144; i++) lens[i] = 8;
            for (int i = 144; i < 256; i++) lens[i] = 9;
            for (int i = 256; i < 280; i++) lens[i] = 7;
            for (int i = 280; i < 288; i++) lens[i] = 8;
            if (!huff_build(lit, lens, 288, 11))
                return false;
            for (int i = 0; i < 30; i++) lens[i] = 5;
            if (!huff_build(dist, lens, 30, 8))
                return false;
        } else {
            refill();
            if (bn < 14)
                return false;
            const int nlen = (int)take(5) + 257, ndist = (int)take(5) + 1, ncode = (int)take(4) + 4;
            if (nlen > 286 || ndist > 30)
                return false;
            unsigned char cl[19] = {0};
            for (int i = 0; i < ncode; i++) {
                refill();
                if (bn < 3)
                    return false;
                cl[CLORD[i]] = (unsigned char)take(3);
            }
            HuffDec clh;
            if (!huff_build(clh, cl, 19, 7))
                return false;
            int i = 0;
            while (i < nlen + ndist) {
                refill();
                const uint16_t e = clh.fast[bb & 127];
                if (!e || (e & 15) > bn)
                    return false;
                take(e & 15);
                const int sym = e >> 4;
                if (sym < 16) {
                    lens[i++] = (unsigned char)sym;
                    continue;
                }
                int rep, val = 0;
                if (bn < 7)
                    return false;
                if (sym == 16) {
                    if (i == 0)
                        return false;
                    val = lens[i - 1];
                    rep = 3 + (int)take(2);
                } else if (sym == 17) {
                    rep = 3 + (int)take(3);
                } else {
                    rep = 11 + (int)take(7);
                }
                if (i + rep > nlen + ndist)
                    return false;
                while (rep--)
                    lens[i++] = (unsigned char)val;
            }
            if (lens[256] == 0)
                return false;
            if (!huff_build(lit, lens, nlen, 11) || !huff_build(dist, lens + nlen, ndist, 8))
                return false;
        }
        for (;;) {
            refill();
            int sym, used;
            uint16_t e = lit.fast[bb & 2047];
            if (e) {
                used = e & 15;
                sym = e >> 4;
            } else {
                sym = huff_slow(lit, bb, bn, &used);
                if (sym < 0)
                    return false;
            }
            if (used > bn)
                return false;
            bb >>= used;
            bn -= used;
            if (sym < 256) {
                if (op >= oend)
                    return false;
                *op++ = (unsigned char)sym;
                // up to three more literals from the same refill (table codes have at most 11 bits); a table entry
                // below 256 << 4 is a literal
                for (int k = 0; k < 3; k++) {
                    e = lit.fast[bb & 2047];
                    if (!e || e >= (256 << 4) || (e & 15) > bn || op >= oend)
                        break;
                    bb >>= (e & 15);
                    bn -= (e & 15);
                    *op++ = (unsigned char)(e >> 4);
                }
                continue;
            }
            if (sym == 256)
                break;
            sym -= 257;
            if (sym >= 29 || LEXT[sym] > bn)
                return false;
            const size_t length = LBASE[sym] + take(LEXT[sym]);
            refill();
            int dsym;
            e = dist.fast[bb & 255];
            if (e) {
                used = e & 15;
                dsym = e >> 4;
            } else {
                dsym = huff_slow(dist, bb, bn, &used);
                if (dsym < 0)
                    return false;
            }
            if (dsym >= 30 || used + DEXT[dsym] > bn)
                return false;
            bb >>= used;
            bn -= used;
            const size_t d = DBASE[dsym] + take(DEXT[dsym]);
            if (d > (size_t)(op - out) || length > (size_t)(oend - op))
                return false;
            const unsigned char *from = op - d;
            if (d >= length) {
                std::memcpy(op, from, length);
                op += length;
            } else {
                for (size_t k = 0; k < length; k++) // overlapping: byte order matters
                    *op++ = *from++;
            }
        }
    }
    if (op != oend)
        return false;
    const uint32_t want = be32(in + in_len - 4);
    return (uint32_t)adler32(adler32(0, Z_NULL, 0), out, (uInt)out_len) == want;
}

struct Raw {
    int W = 0, H = 0, depth = 0, ctype = 0, channels = 0;
    std::vector<unsigned char> pix;     // unfiltered scanlines, H * rowbytes
    std::vector<unsigned char> palette; // RGB triples
    size_t rowbytes = 0;
    void reset_header() { W = H = depth = ctype = channels = 0; rowbytes = 0; }
};

inline int paeth(int a, int b, int c)
{
    // p - a = b - c, p - b = a - c, p - c = (b - c) + (a - c); ties prefer a, then b (PNG spec 9.4)
    const int db = b - c, da = a - c;
    const int pa = std::abs(db), pb = std::abs(da), pc = std::abs(db + da);
    // selects by mask: the comparisons are data dependent and mispredict as branches
    const int mb = -(int)(pb < pa);
    const int r = a ^ ((a ^ b) & mb), pm = pa ^ ((pa ^ pb) & mb);
    const int mc = -(int)(pc < pm);
    return r ^ ((r ^ c) & mc);
}

// Per-thread buffers of the reader (file bytes, gathered IDAT data, the decoded image): decoder threads of
// a batch run reuse them from image to image instead of mapping and unmapping ~1.5 MB per file, which
// serialises the threads of a process on the kernel's address-space lock.
struct Scratch {
    std::vector<unsigned char> file, idat;
    Raw raw;
};
inline Scratch &scratch()
{
    thread_local Scratch s;
    return s;
}

// The returned image lives in the calling thread's scratch: valid until that thread decodes again.
inline const Raw &decode(const std::string &path)
{
    Scratch &sc = scratch();
    std::vector<unsigned char> &f = sc.file;
    slurp_into(path, f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0)
        throw std::runtime_error(path + ": not a PNG file");
    Raw &r = sc.raw;
    r.reset_header();
    r.palette.clear();
    std::vector<unsigned char> &idat = sc.idat;
    idat.clear();
    const unsigned char *z = nullptr; // the IDAT data: in place when the file has one IDAT chunk
    size_t zlen = 0;
    int nidat = 0;
    size_t pos = 8;
    bool have_hdr = false, end = false;
    int interlace = 0;
    while (!end && pos + 12 <= f.size()) {
        const uint32_t len = be32(&f[pos]);
        const char *type = reinterpret_cast<const char *>(&f[pos + 4]);
        if ((size_t)len > f.size() - pos - 12)
            throw std::runtime_error(path + ": truncated chunk");
        const unsigned char *d = &f[pos + 8];
        if (crc32(crc32(0, Z_NULL, 0), &f[pos + 4], len + 4) != be32(d + len))
            throw std::runtime_error(path + ": chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13)
                throw std::runtime_error(path + ": bad IHDR");
            r.W = (int)be32(d);
            r.H = (int)be32(d + 4);
            r.depth = d[8];
            r.ctype = d[9];
            interlace = d[12];
            have_hdr = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            r.palette.assign(d, d + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            if (nidat++ == 0) {
                z = d;
                zlen = len;
            } else {
                if (nidat == 2)
                    idat.assign(z, z + zlen);
                idat.insert(idat.end(), d, d + len);
                z = idat.data();
                zlen = idat.size();
            }
        } else if (!std::memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_hdr || r.W <= 0 || r.H <= 0 || r.W > (1 << 16) || r.H > (1 << 16))
        throw std::runtime_error(path + ": missing or bad header");
    if (interlace)
        throw std::runtime_error(path + ": interlaced PNG is not supported");
    switch (r.ctype) {
    case 0: r.channels = 1; break;
    case 2: r.channels = 3; break;
    case 3: r.channels = 1; break;
    case 4: r.channels = 2; break;
    case 6: r.channels = 4; break;
    default: throw std::runtime_error(path + ": bad colour type");
    }
    const bool depth_ok = r.ctype == 0   ? (r.depth == 1 || r.depth == 2 || r.depth == 4 || r.depth == 8 || r.depth == 16)
                          : r.ctype == 3 ? (r.depth == 1 || r.depth == 2 || r.depth == 4 || r.depth == 8)
                                         : (r.depth == 8 || r.depth == 16);
    if (!depth_ok)
        throw std::runtime_error(path + ": bad bit depth");
    r.rowbytes = ((size_t)r.W * r.channels * r.depth + 7) / 8;
    const size_t bpp = (size_t)(r.channels * r.depth + 7) / 8; // filter distance, at least 1
    std::vector<unsigned char> &raw = r.pix; // filtered scanlines first, unfiltered in place below
    raw.resize((r.rowbytes + 1) * (size_t)r.H);
    if (nidat > 1) {
        z = idat.data(); // (the vector may have moved while it grew)
        zlen = idat.size();
    }
    if (!z || !inflate_zlib_exact(z, zlen, raw.data(), raw.size())) {
        uLongf out_len = (uLongf)raw.size();
        const int zrc = z ? uncompress(raw.data(), &out_len, z, (uLong)zlen) : Z_DATA_ERROR;
        if (zrc != Z_OK || out_len != raw.size())
            throw std::runtime_error(path + ": image data does not inflate to the declared size");
    }
    // Unfiltered in place: row y ends up at raw[y * rowbytes], always at or below the bytes still to be read
    // (its own filtered bytes start at y * (rowbytes + 1) + 1).
    const std::vector<unsigned char> zero(r.rowbytes, 0);
    for (int y = 0; y < r.H; y++) {
        const unsigned char *src = &raw[(size_t)y * (r.rowbytes + 1)];
        unsigned char *cur = &raw[(size_t)y * r.rowbytes];
        const unsigned char *up = y ? cur - r.rowbytes : zero.data();
        const int ft = src[0];
        if (ft > 4)
            throw std::runtime_error(path + ": bad filter type");
        const unsigned char *in = src + 1;
        const size_t n = r.rowbytes, head = bpp < n ? bpp : n;
        switch (ft) {
        case 0:
            std::memmove(cur, in, n);
            break;
        case 1:
            for (size_t i = 0; i < head; i++)
                cur[i] = in[i];
            for (size_t i = head; i < n; i++)
                cur[i] = (unsigned char)(in[i] + cur[i - bpp]);
            break;
        case 2:
            for (size_t i = 0; i < n; i++)
                cur[i] = (unsigned char)(in[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < head; i++)
                cur[i] = (unsigned char)(in[i] + (up[i] >> 1));
            for (size_t i = head; i < n; i++)
                cur[i] = (unsigned char)(in[i] + ((cur[i - bpp] + up[i]) >> 1));
            break;
        default:
            for (size_t i = 0; i < head; i++)
                cur[i] = (unsigned char)(in[i] + paeth(0, up[i], 0));
            if (bpp == 1) { // 8-bit gray: left and upper-left neighbours carried in registers
                int a = cur[0], c = up[0];
                for (size_t i = 1; i < n; i++) {
                    const int b = up[i];
                    a = (in[i] + paeth(a, b, c)) & 0xff;
                    cur[i] = (unsigned char)a;
                    c = b;
                }
                break;
            }
            for (size_t i = head; i < n; i++)
                cur[i] = (unsigned char)(in[i] + paeth(cur[i - bpp], up[i], up[i - bpp]));
            break;
        }
    }
    raw.resize(r.rowbytes * (size_t)r.H);
    return r;
}

inline unsigned sample(const Raw &r, const unsigned char *row, size_t idx)
{
    if (r.depth == 8)
        return row[idx];
    if (r.depth == 16)
        return ((unsigned)row[2 * idx] << 8) | row[2 * idx + 1];
    const size_t bit = idx * (size_t)r.depth;
    return (row[bit >> 3] >> (8 - r.depth - (bit & 7))) & ((1u << r.depth) - 1);
}

// cv::imread(path, CV_LOAD_IMAGE_GRAYSCALE): always 8-bit, one channel.
inline Mat read_gray8(const std::string &path)
{
    const Raw &r = decode(path);
    Mat m(r.H, r.W, MAT_8U);
    for (int y = 0; y < r.H; y++) {
        const unsigned char *row = &r.pix[(size_t)y * r.rowbytes];
        unsigned char *dst = m.data + (size_t)y * m.step;
        if (r.ctype == 0 && r.depth == 8) { // the KITTI / Tsukuba inputs: already 8-bit gray
            std::memcpy(dst, row, (size_t)r.W);
            continue;
        }
        for (int x = 0; x < r.W; x++) {
            unsigned R, G, B;
            if (r.ctype == 3) {
                const unsigned i = sample(r, row, (size_t)x);
                if (3 * i + 2 >= r.palette.size())
                    throw std::runtime_error(path + ": palette index out of range");
                R = r.palette[3 * i];
                G = r.palette[3 * i + 1];
                B = r.palette[3 * i + 2];
            } else {
                const size_t base = (size_t)x * r.channels;
                unsigned s[3];
                const int nc = r.channels >= 3 ? 3 : 1;
                for (int c = 0; c < nc; c++) {
                    const unsigned v = sample(r, row, base + c);
                    // to 8 bits: 16-bit keeps the high byte, sub-byte depths scale to 0..255
                    s[c] = r.depth == 16 ? v >> 8 : (r.depth == 8 ? v : v * 255u / ((1u << r.depth) - 1));
                }
                if (nc == 1) {
                    dst[x] = (unsigned char)s[0];
                    continue;
                }
                R = s[0];
                G = s[1];
                B = s[2];
            }
            dst[x] = (unsigned char)((R * 4899u + G * 9617u + B * 1868u + 8192u) >> 14);
        }
    }
    return m;
}

// 16-bit gray file (disparity maps, value = 256 * d) as CV_16U; 8-bit gray is widened unchanged.
inline Mat read_gray16(const std::string &path)
{
    const Raw &r = decode(path);
    if (r.ctype != 0 || (r.depth != 16 && r.depth != 8))
        throw std::runtime_error(path + ": want an 8- or 16-bit gray PNG");
    Mat m(r.H, r.W, MAT_16U);
    for (int y = 0; y < r.H; y++) {
        const unsigned char *row = &r.pix[(size_t)y * r.rowbytes];
        for (int x = 0; x < r.W; x++)
            m.at<uint16_t>(y, x) = (uint16_t)sample(r, row, (size_t)x);
    }
    return m;
}

// ---------------------------------------------------------------------------
// Fast zlib stream for the default writer: one dynamic-Huffman deflate block of literals only (no
// match search).  On Sub-filtered disparity and gray images this is as small as zlib's Z_RLE /
// Z_HUFFMAN_ONLY output (the filter has already removed what a match search finds) at several times
// the speed; any inflater reads it.  Little-endian hosts (the bit buffer is stored with memcpy).
// ---------------------------------------------------------------------------

// Code lengths (<= maxlen, Kraft sum exactly 1) of a Huffman code for the symbols with freq > 0;
// needs at least two of them.  Lengths come from the Huffman tree; when the tree is deeper than
// maxlen the length counts are folded down to maxlen and repaired one leaf at a time, then the
// multiset of lengths is dealt to the symbols from the rarest (longest) to the most frequent.
inline void huffman_lengths(const uint32_t *freq, int nsym, int maxlen, unsigned char *len)
{
    struct Node {
        uint64_t w;
        int parent;
    };
    std::vector<Node> nodes;
    std::vector<int> used;
    for (int i = 0; i < nsym; i++) {
        len[i] = 0;
        if (freq[i]) {
            used.push_back(i);
            nodes.push_back({freq[i], -1});
        }
    }
    const int n = (int)used.size();
    if (n < 2)
        throw std::logic_error("huffman_lengths: needs two symbols");
    // two-queue merge over the leaves sorted by weight (ties by symbol: deterministic files)
    std::vector<int> order(n);
    for (int i = 0; i < n; i++)
        order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return nodes[a].w != nodes[b].w ? nodes[a].w < nodes[b].w : a < b; });
    size_t li = 0, ii = (size_t)n; // next unmerged leaf (in `order`), next unmerged internal node
    auto take = [&]() {
        const bool leaf = li < (size_t)n && (ii >= nodes.size() || nodes[order[li]].w <= nodes[ii].w);
        return leaf ? order[li++] : (int)ii++;
    };
    for (int k = 0; k < n - 1; k++) {
        const int a = take(), b = take();
        nodes.push_back({nodes[a].w + nodes[b].w, -1});
        nodes[a].parent = nodes[b].parent = (int)nodes.size() - 1;
    }
    std::vector<int> count(64, 0);
    for (int i = 0; i < n; i++) {
        int d = 0;
        for (int v = i; nodes[v].parent >= 0; v = nodes[v].parent)
            d++;
        count[d < 63 ? d : 63]++;
    }
    for (int d = maxlen + 1; d < 64; d++) {
        count[maxlen] += count[d];
        count[d] = 0;
    }
    uint64_t total = 0;
    for (int d = 1; d <= maxlen; d++)
        total += (uint64_t)count[d] << (maxlen - d);
    while (total > ((uint64_t)1 << maxlen)) { // over-subscribed: drop one leaf at maxlen, split one above it
        count[maxlen]--;
        for (int d = maxlen - 1; d > 0; d--)
            if (count[d]) {
                count[d]--;
                count[d + 1] += 2;
                break;
            }
        total--;
    }
    int d = maxlen;
    for (int i = 0; i < n; i++) { // rarest first
        while (count[d] == 0)
            d--;
        len[used[order[i]]] = (unsigned char)d;
        count[d]--;
    }
}

// Canonical deflate codes (RFC 1951 3.2.2) for the lengths, bit-reversed for the LSB-first bit stream.
inline void canonical_codes(const unsigned char *len, int nsym, uint16_t *code)
{
    int bl_count[16] = {0}, next[16] = {0};
    for (int i = 0; i < nsym; i++)
        bl_count[len[i]]++;
    bl_count[0] = 0;
    for (int b = 1, c = 0; b < 16; b++) {
        c = (c + bl_count[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < nsym; i++) {
        unsigned v = len[i] ? (unsigned)next[len[i]]++ : 0u, r = 0;
        for (int b = 0; b < len[i]; b++)
            r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        code[i] = (uint16_t)r;
    }
}

inline std::vector<unsigned char> zlib_huffman_only(const unsigned char *src, size_t n)
{
    uint32_t freq[257] = {0};
    for (size_t i = 0; i < n; i++)
        freq[src[i]]++;
    freq[256] = 1; // end of block
    if (n == 0)
        freq[0] = 1; // a second symbol keeps the code complete
    unsigned char len[257];
    uint16_t code[257];
    huffman_lengths(freq, 257, 15, len);
    canonical_codes(len, 257, code);
    uint64_t bits = 0;
    for (int i = 0; i < 257; i++)
        bits += (uint64_t)freq[i] * len[i];
    std::vector<unsigned char> out(2 + 160 + (size_t)(bits / 8) + 16 + 4);
    unsigned char *p = out.data();
    *p++ = 0x78; // deflate, 32 KiB window
    *p++ = 0x01; // fastest; (0x7801 % 31 == 0)
    uint64_t acc = 0;
    int na = 0;
    auto put = [&](uint32_t v, int nb) {
        acc |= (uint64_t)v << na;
        na += nb;
        if (na >= 32) {
            const uint32_t w = (uint32_t)acc;
            std::memcpy(p, &w, 4);
            p += 4;
            acc >>= 32;
            na -= 32;
        }
    };
    put(1, 1);  // BFINAL
    put(2, 2);  // dynamic Huffman
    put(0, 5);  // HLIT: 257 literal/length codes
    put(0, 5);  // HDIST: 1 distance code (of zero bits: no distances are used)
    put(15, 4); // HCLEN: all 19 code-length code lengths
    // code-length alphabet: symbols 0..15 with 4 bits each (a complete code), 16/17/18 unused
    static const int cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 19; i++)
        put(cl_order[i] < 16 ? 4 : 0, 3);
    auto put_len = [&](unsigned l) { // canonical 4-bit code of symbol l is l itself, sent MSB first
        put(((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3), 4);
    };
    for (int i = 0; i < 257; i++)
        put_len(len[i]);
    put_len(0); // the one distance code
    for (size_t i = 0; i < n; i++)
        put(code[src[i]], len[src[i]]);
    put(code[256], len[256]);
    while (na > 0) { // flush whole bytes
        *p++ = (unsigned char)acc;
        acc >>= 8;
        na -= 8;
    }
    const uint32_t ad = (uint32_t)adler32(adler32(0, Z_NULL, 0), src, (uInt)n);
    *p++ = (unsigned char)(ad >> 24);
    *p++ = (unsigned char)(ad >> 16);
    *p++ = (unsigned char)(ad >> 8);
    *p++ = (unsigned char)ad;
    out.resize((size_t)(p - out.data()));
    return out;
}

inline void chunk(std::vector<unsigned char> &out, const char *type, const unsigned char *d, size_t len)
{
    put32(out, (uint32_t)len);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (len)
        out.insert(out.end(), d, d + len);
    put32(out, (uint32_t)crc32(crc32(0, Z_NULL, 0), &out[at], (uInt)(len + 4)));
}

// cv::imwrite(path, mat) for CV_8U / CV_16U single channel.
//   level < 0 (default): the speed setting.  OpenCV's PNG encoder, when no compression parameter is passed (as in
//     sgm_single/main.cc:55-56 and sgm_cnn/main.cc:49-50), takes the Sub filter on every row with Z_BEST_SPEED and
//     the Z_RLE strategy ("tune parameters for speed", grfmt_png.cpp of the 2.4 / 3.x library); here the same Sub
//     filter feeds zlib_huffman_only above (same size within 1 %, about 5x faster than zlib's Z_RLE).
//   level 0..9: per-row filter chosen by the usual minimum-sum-of-absolute-differences heuristic and that zlib
//     level (the mains that pass compression_params ask for 9, dense_stereo/main.cc:70-72,109).
// The pixel values read back are the same either way; only the file size and the time differ.
template <class MatT> inline void write_gray(const std::string &path, const MatT &m, int level = -1)
{
    if (m.type() != MAT_8U && m.type() != MAT_16U)
        throw std::runtime_error("png::write_gray: 8- or 16-bit single channel only");
    const int bps = m.type() == MAT_16U ? 2 : 1;
    const size_t rowbytes = (size_t)m.cols * bps;
    std::vector<unsigned char> prev(rowbytes, 0), cur(rowbytes), best(rowbytes), cand(rowbytes), raw;
    raw.reserve((rowbytes + 1) * (size_t)m.rows);
    if (level < 0)
        raw.resize((rowbytes + 1) * (size_t)m.rows);
    for (int y = 0; y < m.rows; y++) {
        const unsigned char *src = m.data + (size_t)y * (size_t)m.step;
        if (level < 0) { // Sub filter on every row, straight from the source pixels
            unsigned char *o = &raw[(size_t)y * (rowbytes + 1)];
            *o++ = 1;
            if (bps == 2) {
                const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
                unsigned ph = 0, pl = 0;
                for (int x = 0; x < m.cols; x++) { // big-endian samples
                    const unsigned h = s16[x] >> 8, l = s16[x] & 255u;
                    o[2 * x] = (unsigned char)(h - ph);
                    o[2 * x + 1] = (unsigned char)(l - pl);
                    ph = h;
                    pl = l;
                }
            } else {
                unsigned pv = 0;
                for (int x = 0; x < m.cols; x++) {
                    o[x] = (unsigned char)(src[x] - pv);
                    pv = src[x];
                }
            }
            continue;
        }
        if (bps == 2)
            for (int x = 0; x < m.cols; x++) { // big-endian samples
                const uint16_t v = reinterpret_cast<const uint16_t *>(src)[x];
                cur[2 * x] = (unsigned char)(v >> 8);
                cur[2 * x + 1] = (unsigned char)v;
            }
        else
            std::memcpy(cur.data(), src, rowbytes);
        long best_sum = -1;
        int best_ft = 0;
        for (int ft = 0; ft < 5; ft++) {
            long sum = 0;
            for (size_t i = 0; i < rowbytes; i++) {
                const int a = i >= (size_t)bps ? cur[i - bps] : 0, b = prev[i], c = i >= (size_t)bps ? prev[i - bps] : 0;
                int p = 0;
                switch (ft) {
                case 1: p = a; break;
                case 2: p = b; break;
                case 3: p = (a + b) >> 1; break;
                case 4: p = paeth(a, b, c); break;
                default: break;
                }
                const unsigned char v = (unsigned char)(cur[i] - p);
                cand[i] = v;
                sum += v < 128 ? v : 256 - v;
            }
            if (best_sum < 0 || sum < best_sum) {
                best_sum = sum;
                best_ft = ft;
                best.swap(cand);
            }
        }
        raw.push_back((unsigned char)best_ft);
        raw.insert(raw.end(), best.begin(), best.end());
        prev = cur;
    }
    uLongf zlen = 0;
    std::vector<unsigned char> z;
    if (level < 0) {
        z = zlib_huffman_only(raw.data(), raw.size());
        zlen = (uLongf)z.size();
    } else {
        zlen = compressBound((uLong)raw.size());
        z.resize(zlen);
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
            throw std::runtime_error("png::write_gray: deflateInit2 failed");
        zs.next_in = raw.data();
        zs.avail_in = (uInt)raw.size();
        zs.next_out = z.data();
        zs.avail_out = (uInt)z.size();
        const int rc = deflate(&zs, Z_FINISH);
        zlen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END)
            throw std::runtime_error("png::write_gray: deflate failed");
    }
    std::vector<unsigned char> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<unsigned char> hdr;
    put32(hdr, (uint32_t)m.cols);
    put32(hdr, (uint32_t)m.rows);
    hdr.push_back((unsigned char)(8 * bps));
    hdr.push_back(0); // gray
    hdr.push_back(0);
    hdr.push_back(0);
    hdr.push_back(0);
    chunk(out, "IHDR", hdr.data(), hdr.size());
    chunk(out, "IDAT", z.data(), zlen);
    chunk(out, "IEND", nullptr, 0);
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f)
        throw std::runtime_error("cannot write " + path);
    const size_t n = std::fwrite(out.data(), 1, out.size(), f);
    if (std::fclose(f) != 0 || n != out.size())
        throw std::runtime_error("short write on " + path);
}

} // namespace png
} // namespace sgmx
#endif
