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

inline std::vector<unsigned char> slurp(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f)
        throw std::runtime_error("cannot open " + path);
    std::vector<unsigned char> buf;
    unsigned char tmp[1 << 16];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0)
        buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    return buf;
}

struct Raw {
    int W = 0, H = 0, depth = 0, ctype = 0, channels = 0;
    std::vector<unsigned char> pix;     // unfiltered scanlines, H * rowbytes
    std::vector<unsigned char> palette; // RGB triples
    size_t rowbytes = 0;
};

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

inline Raw decode(const std::string &path)
{
    const std::vector<unsigned char> f = slurp(path);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (f.size() < 8 + 25 || std::memcmp(f.data(), sig, 8) != 0)
        throw std::runtime_error(path + ": not a PNG file");
    Raw r;
    std::vector<unsigned char> idat;
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
            idat.insert(idat.end(), d, d + len);
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
    std::vector<unsigned char> raw((r.rowbytes + 1) * (size_t)r.H);
    uLongf out_len = (uLongf)raw.size();
    const int zrc = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zrc != Z_OK || out_len != raw.size())
        throw std::runtime_error(path + ": image data does not inflate to the declared size");
    r.pix.resize(r.rowbytes * (size_t)r.H);
    const std::vector<unsigned char> zero(r.rowbytes, 0);
    for (int y = 0; y < r.H; y++) {
        const unsigned char *src = &raw[(size_t)y * (r.rowbytes + 1)];
        unsigned char *cur = &r.pix[(size_t)y * r.rowbytes];
        const unsigned char *up = y ? cur - r.rowbytes : zero.data();
        const int ft = src[0];
        if (ft > 4)
            throw std::runtime_error(path + ": bad filter type");
        const unsigned char *in = src + 1;
        const size_t n = r.rowbytes, head = bpp < n ? bpp : n;
        switch (ft) {
        case 0:
            std::memcpy(cur, in, n);
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
            for (size_t i = head; i < n; i++)
                cur[i] = (unsigned char)(in[i] + paeth(cur[i - bpp], up[i], up[i - bpp]));
            break;
        }
    }
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
    const Raw r = decode(path);
    Mat m(r.H, r.W, MAT_8U);
    for (int y = 0; y < r.H; y++) {
        const unsigned char *row = &r.pix[(size_t)y * r.rowbytes];
        unsigned char *dst = m.data + (size_t)y * m.step;
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
    const Raw r = decode(path);
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
//   level < 0 (default): what OpenCV's PNG encoder does when no compression parameter is passed, as in
//     sgm_single/main.cc:55-56 and sgm_cnn/main.cc:49-50 -- the Sub filter on every row, Z_BEST_SPEED and the
//     Z_RLE strategy ("tune parameters for speed", grfmt_png.cpp of the 2.4 / 3.x library).
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
    for (int y = 0; y < m.rows; y++) {
        const unsigned char *src = m.data + (size_t)y * (size_t)m.step;
        if (bps == 2)
            for (int x = 0; x < m.cols; x++) { // big-endian samples
                const uint16_t v = reinterpret_cast<const uint16_t *>(src)[x];
                cur[2 * x] = (unsigned char)(v >> 8);
                cur[2 * x + 1] = (unsigned char)v;
            }
        else
            std::memcpy(cur.data(), src, rowbytes);
        if (level < 0) {
            raw.push_back(1); // Sub
            const size_t at = raw.size();
            raw.resize(at + rowbytes);
            unsigned char *o = &raw[at];
            for (size_t i = 0; i < rowbytes && i < (size_t)bps; i++)
                o[i] = cur[i];
            for (size_t i = bps; i < rowbytes; i++)
                o[i] = (unsigned char)(cur[i] - cur[i - bps]);
            continue;
        }
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
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zlen);
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, level < 0 ? Z_BEST_SPEED : level, Z_DEFLATED, 15, 8, level < 0 ? Z_RLE : Z_DEFAULT_STRATEGY) != Z_OK)
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
