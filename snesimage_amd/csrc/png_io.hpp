// snesimage_amd/csrc/png_io.hpp — PNG in, PNG out for the headless driver (host only, zlib for inflate/deflate).
//
// Input: what `image::open(path)?.into_rgba8()` (src/lib.rs:836) yields for a PNG: every colour type and bit depth,
// interlaced or not, expanded to RGBA8 — grey replicated into r,g,b; palette entries looked up; a tRNS chunk becomes
// alpha; samples below 8 bits scaled to the full range; 16-bit samples reduced with round(c * 255 / 65535) =
// (c + 128) / 257 (image 0.25's u16 -> u8 conversion).  Gamma and colour-profile chunks are ignored, as the
// reference's decoder ignores them.  Other container formats of the `image` crate are out of scope.
// Output: RGBA8, filter 0 on every row — used for the preview that stands in for the SDL window (src/lib.rs:937-960).
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

namespace snes_png {

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }
inline bool is_png(const uint8_t *d, size_t n) { static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10}; return n >= 8 && memcmp(d, sig, 8) == 0; }

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Undo the row filters of one (sub)image in place: `rows` scanlines of `stride` bytes, each preceded by its filter byte.
inline bool unfilter(uint8_t *p, size_t rows, size_t stride, size_t bpp, std::string &err) {
    std::vector<uint8_t> zero(stride, 0);
    const uint8_t *prev = zero.data();
    for (size_t y = 0; y < rows; y++) {
        const uint8_t ft = p[0];
        uint8_t *cur = p + 1;
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + (((i >= bpp ? cur[i - bpp] : 0) + prev[i]) >> 1)); break;
        case 4: for (size_t i = 0; i < stride; i++) cur[i] = (uint8_t)(cur[i] + paeth(i >= bpp ? cur[i - bpp] : 0, prev[i], i >= bpp ? prev[i - bpp] : 0)); break;
        default: err = "invalid PNG filter type"; return false;
        }
        prev = cur;
        p += stride + 1;
    }
    return true;
}

struct Header { uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0; };

inline int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }

// sample s of a row (depth bits each), most significant bits first
inline uint32_t sample(const uint8_t *row, size_t s, int depth) {
    if (depth == 8) return row[s];
    if (depth == 16) return ((uint32_t)row[2 * s] << 8) | row[2 * s + 1];
    const size_t bit = s * (size_t)depth;
    return (row[bit >> 3] >> (8 - depth - (int)(bit & 7))) & ((1u << depth) - 1u);
}
inline uint8_t to8(uint32_t v, int depth) { // full-range scaling of a sample to 8 bits
    switch (depth) {
    case 1: return v ? 255 : 0;
    case 2: return (uint8_t)(v * 85u);
    case 4: return (uint8_t)(v * 17u);
    case 8: return (uint8_t)v;
    default: return (uint8_t)((v + 128u) / 257u);
    }
}

// one decoded row of `n` pixels -> RGBA8 written at out + 4*(x0 + i*dx)
inline void expand_row(const Header &H, const uint8_t *row, uint32_t n, const std::vector<uint8_t> &plte, const std::vector<uint8_t> &trns, uint8_t *out, uint32_t x0, uint32_t dx) {
    const int d = H.depth;
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *o = out + 4 * ((size_t)x0 + (size_t)i * dx);
        switch (H.ctype) {
        case 0: { const uint32_t g = sample(row, i, d); const uint8_t g8 = to8(g, d); o[0] = o[1] = o[2] = g8;
                  o[3] = (trns.size() >= 2 && g == ((((uint32_t)trns[0] << 8) | trns[1]) & ((1u << d) - 1u))) ? 0 : 255; break; }
        case 2: { const uint32_t r = sample(row, 3 * (size_t)i, d), g = sample(row, 3 * (size_t)i + 1, d), b = sample(row, 3 * (size_t)i + 2, d);
                  o[0] = to8(r, d); o[1] = to8(g, d); o[2] = to8(b, d);
                  const uint32_t m = (d == 16) ? 0xffffu : 0xffu;
                  o[3] = (trns.size() >= 6 && r == ((((uint32_t)trns[0] << 8) | trns[1]) & m) && g == ((((uint32_t)trns[2] << 8) | trns[3]) & m) && b == ((((uint32_t)trns[4] << 8) | trns[5]) & m)) ? 0 : 255; break; }
        case 3: { const uint32_t ix = sample(row, i, d);
                  if (3 * (size_t)ix + 2 < plte.size()) { o[0] = plte[3 * ix]; o[1] = plte[3 * ix + 1]; o[2] = plte[3 * ix + 2]; } else { o[0] = o[1] = o[2] = 0; }
                  o[3] = ix < trns.size() ? trns[ix] : 255; break; }
        case 4: { const uint8_t g8 = to8(sample(row, 2 * (size_t)i, d), d); o[0] = o[1] = o[2] = g8; o[3] = to8(sample(row, 2 * (size_t)i + 1, d), d); break; }
        default: { o[0] = to8(sample(row, 4 * (size_t)i, d), d); o[1] = to8(sample(row, 4 * (size_t)i + 1, d), d); o[2] = to8(sample(row, 4 * (size_t)i + 2, d), d);
                   o[3] = to8(sample(row, 4 * (size_t)i + 3, d), d); break; }
        }
    }
}

inline bool decode(const uint8_t *data, size_t n, uint32_t &w, uint32_t &h, std::vector<uint8_t> &rgba, std::string &err) {
    if (!is_png(data, n)) { err = "not a PNG file"; return false; }
    Header H; bool have_ihdr = false, have_iend = false;
    std::vector<uint8_t> plte, trns, idat;
    size_t pos = 8;
    while (pos + 12 <= n && !have_iend) {
        const uint32_t len = be32(data + pos);
        if ((size_t)len > n - pos - 12) { err = "truncated PNG chunk"; return false; }
        const uint8_t *type = data + pos + 4, *body = data + pos + 8;
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) != be32(body + len)) { err = "PNG chunk CRC mismatch"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (len != 13) { err = "bad IHDR"; return false; }
            H.w = be32(body); H.h = be32(body + 4); H.depth = body[8]; H.ctype = body[9]; H.interlace = body[12];
            if (body[10] != 0 || body[11] != 0 || H.interlace > 1) { err = "unsupported PNG compression/filter/interlace method"; return false; }
            const bool ok_depth = (H.ctype == 0 && (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8 || H.depth == 16)) ||
                                  (H.ctype == 3 && (H.depth == 1 || H.depth == 2 || H.depth == 4 || H.depth == 8)) ||
                                  ((H.ctype == 2 || H.ctype == 4 || H.ctype == 6) && (H.depth == 8 || H.depth == 16));
            if (!ok_depth || H.w == 0 || H.h == 0 || H.w > 16384 || H.h > 16384) { err = "unsupported PNG header"; return false; }
            have_ihdr = true;
        } else if (!have_ihdr) { err = "PNG does not start with IHDR"; return false; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) have_iend = true;
        else if (!(type[0] & 0x20)) { err = "unknown critical PNG chunk"; return false; }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend || idat.empty()) { err = "incomplete PNG (IHDR/IDAT/IEND)"; return false; }
    if (H.ctype == 3 && plte.empty()) { err = "palette PNG without PLTE"; return false; }
    const int ch = channels_of(H.ctype);
    const size_t bits = (size_t)ch * H.depth, bpp = bits >= 8 ? bits / 8 : 1;
    auto stride_of = [&](uint32_t pw) { return ((size_t)pw * bits + 7) / 8; };
    // Adam7 passes (x0, y0, dx, dy); a non-interlaced image is the single pass (0,0,1,1)
    static const uint32_t a7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    const int npass = H.interlace ? 7 : 1;
    size_t raw = 0;
    uint32_t pw[7], ph[7];
    for (int p = 0; p < npass; p++) {
        const uint32_t x0 = H.interlace ? a7[p][0] : 0, y0 = H.interlace ? a7[p][1] : 0, dx = H.interlace ? a7[p][2] : 1, dy = H.interlace ? a7[p][3] : 1;
        pw[p] = H.w > x0 ? (H.w - x0 + dx - 1) / dx : 0; ph[p] = H.h > y0 ? (H.h - y0 + dy - 1) / dy : 0;
        if (pw[p] && ph[p]) raw += (stride_of(pw[p]) + 1) * ph[p];
    }
    std::vector<uint8_t> buf(raw);
    uLongf got = (uLongf)raw;
    const int zr = uncompress(buf.data(), &got, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || (size_t)got != raw) { err = "PNG image data does not inflate to the declared size"; return false; }
    w = H.w; h = H.h;
    rgba.assign((size_t)w * h * 4, 0);
    uint8_t *p = buf.data();
    for (int ps = 0; ps < npass; ps++) {
        if (!pw[ps] || !ph[ps]) continue;
        const uint32_t x0 = H.interlace ? a7[ps][0] : 0, y0 = H.interlace ? a7[ps][1] : 0, dx = H.interlace ? a7[ps][2] : 1, dy = H.interlace ? a7[ps][3] : 1;
        const size_t stride = stride_of(pw[ps]);
        if (!unfilter(p, ph[ps], stride, bpp, err)) return false;
        for (uint32_t r = 0; r < ph[ps]; r++) expand_row(H, p + (size_t)r * (stride + 1) + 1, pw[ps], plte, trns, rgba.data() + (size_t)(y0 + r * dy) * w * 4, x0, dx);
        p += (stride + 1) * ph[ps];
    }
    return true;
}

inline void put_chunk(std::vector<uint8_t> &out, const char *type, const uint8_t *body, size_t len) {
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    out.insert(out.end(), l, l + 4);
    const size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), body, body + len);
    const uint32_t c = (uint32_t)crc32(crc32(0L, Z_NULL, 0), out.data() + at, (uInt)(4 + len));
    const uint8_t cb[4] = {(uint8_t)(c >> 24), (uint8_t)(c >> 16), (uint8_t)(c >> 8), (uint8_t)c};
    out.insert(out.end(), cb, cb + 4);
}

inline bool encode_rgba(uint32_t w, uint32_t h, const uint8_t *rgba, std::vector<uint8_t> &out) {
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    out.assign(sig, sig + 8);
    const uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0};
    put_chunk(out, "IHDR", ihdr, 13);
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
    for (uint32_t y = 0; y < h; y++) { raw[(size_t)y * (w * 4 + 1)] = 0; memcpy(&raw[(size_t)y * (w * 4 + 1) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4); }
    uLongf cap = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(cap);
    if (compress2(z.data(), &cap, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    put_chunk(out, "IDAT", z.data(), cap);
    put_chunk(out, "IEND", nullptr, 0);
    return true;
}

} // namespace snes_png
