// Texture and frame files: TGA and PNG decode to RGBA8, TGA and PNG encode from RGBA8.
// Stands in for SOIL_load_OGL_texture(..., SOIL_LOAD_AUTO, ..., SOIL_FLAG_INVERT_Y) of
// "OpenGL Raytracer/main.cpp":391-398 on the way in, and gives the presentation tail a file to
// write on the way out.  Self-contained (its own inflate; CRC-32 and Adler-32 for the writer).
#include "Image.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rth {

namespace {

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    Reader(const uint8_t *b, int64_t n) : p(b), end(b + n) {}
    int64_t left() const { return end - p; }
    uint8_t u8() {
        if (p >= end) { ok = false; return 0; }
        return *p++;
    }
    uint16_t le16() { uint16_t a = u8(); return static_cast<uint16_t>(a | (u8() << 8)); }
    uint32_t be32() {
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) v = (v << 8) | u8();
        return v;
    }
    void skip(int64_t n) {
        if (n < 0 || n > left()) { ok = false; p = end; return; }
        p += n;
    }
};

// ------------------------------------------------------------------------------------------
// TGA.  Pixel interpretation follows the decoder the reference links (SOIL's stb_image_aug.c,
// tga_load): 8 bits = grey, 16 bits = grey + alpha, 24 = BGR, 32 = BGRA; a colour-mapped image
// takes its depth from the palette; the right-to-left bit of the descriptor is ignored.
// ------------------------------------------------------------------------------------------
bool decode_tga(const uint8_t *bytes, int64_t len, Image &img, std::string &err) {
    Reader r(bytes, len);
    const int id_len = r.u8();
    const int indexed = r.u8();
    int type = r.u8();
    const int pal_start = r.le16();
    const int pal_len = r.le16();
    const int pal_bits = r.u8();
    r.le16();
    r.le16();
    const int w = r.le16(), h = r.le16();
    int bpp = r.u8();
    const int desc = r.u8();
    if (!r.ok) { err = "TGA: truncated header"; return false; }
    bool rle = false;
    if (type >= 8) { type -= 8; rle = true; }
    if (w < 1 || h < 1 || type < 1 || type > 3 || (bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32)) {
        err = "TGA: unsupported header";
        return false;
    }
    if (indexed) bpp = pal_bits;
    if (bpp != 8 && bpp != 16 && bpp != 24 && bpp != 32) { err = "TGA: unsupported palette depth"; return false; }
    const int bytes_pp = bpp / 8;
    r.skip(id_len);
    const uint8_t *palette = nullptr;
    if (indexed) {
        r.skip(pal_start);
        palette = r.p;
        r.skip(static_cast<int64_t>(pal_len) * bytes_pp);
        if (pal_len < 1) { err = "TGA: empty palette"; return false; }
    }
    if (!r.ok) { err = "TGA: truncated palette"; return false; }

    img.w = w;
    img.h = h;
    img.rgba.assign(static_cast<size_t>(w) * h * 4, 0);
    uint8_t px[4] = {0, 0, 0, 0};
    int run = 0;
    bool repeating = false, need = true;
    for (int64_t i = 0; i < static_cast<int64_t>(w) * h; i++) {
        if (rle) {
            if (run == 0) {
                const int cmd = r.u8();
                run = 1 + (cmd & 127);
                repeating = (cmd >> 7) != 0;
                need = true;
            } else if (!repeating) {
                need = true;
            }
        } else {
            need = true;
        }
        if (need) {
            uint8_t raw[4] = {0, 0, 0, 0};
            if (indexed) {
                int idx = r.u8();
                if (idx >= pal_len) idx = 0;
                std::memcpy(raw, palette + static_cast<size_t>(idx) * bytes_pp, bytes_pp);
            } else {
                for (int j = 0; j < bytes_pp; j++) raw[j] = r.u8();
            }
            switch (bpp) {
                case 8:  px[0] = px[1] = px[2] = raw[0]; px[3] = 255; break;
                case 16: px[0] = px[1] = px[2] = raw[0]; px[3] = raw[1]; break;
                case 24: px[0] = raw[2]; px[1] = raw[1]; px[2] = raw[0]; px[3] = 255; break;
                default: px[0] = raw[2]; px[1] = raw[1]; px[2] = raw[0]; px[3] = raw[3]; break;
            }
            need = false;
        }
        std::memcpy(&img.rgba[static_cast<size_t>(i) * 4], px, 4);
        run--;
        if (!r.ok) { err = "TGA: truncated pixel data"; return false; }
    }
    // Stored bottom-up unless descriptor bit 5 is set; Image is top-down.
    if (((desc >> 5) & 1) == 0) img.flip_rows();
    return true;
}

// ------------------------------------------------------------------------------------------
// inflate (RFC 1951) inside a zlib wrapper (RFC 1950)
// ------------------------------------------------------------------------------------------
struct Huffman {
    uint16_t count[16];
    uint16_t symbol[288];
    bool build(const uint8_t *lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int l = 1; l < 16; l++) {
            left = (left << 1) - count[l];
            if (left < 0) return false;  // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = static_cast<uint16_t>(offs[l] + count[l]);
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
        return true;
    }
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t bits = 0;
    int nbits = 0;
    bool ok = true;
    int get(int n) {
        while (nbits < n) {
            if (p >= end) { ok = false; return 0; }
            bits |= static_cast<uint32_t>(*p++) << nbits;
            nbits += 8;
        }
        const int v = static_cast<int>(bits & ((1u << n) - 1u));
        bits >>= n;
        nbits -= n;
        return v;
    }
    int decode(const Huffman &h) {
        int code = 0, first = 0, index = 0;
        for (int l = 1; l < 16; l++) {
            code |= get(1);
            if (!ok) return -1;
            const int c = h.count[l];
            if (code - c < first) return h.symbol[index + (code - first)];
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const uint8_t *src, int64_t len, std::vector<uint8_t> &out, size_t expect, std::string &err) {
    if (len < 2 || (src[0] & 15) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 32)) {
        err = "PNG: bad zlib header";
        return false;
    }
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                          35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                           1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                           9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

    BitReader br{src + 2, src + len};
    out.clear();
    out.reserve(expect);
    Huffman lit, dist;
    int last;
    do {
        last = br.get(1);
        const int type = br.get(2);
        if (!br.ok) break;
        if (type == 0) {
            br.bits = 0;
            br.nbits = 0;
            if (br.end - br.p < 4) { br.ok = false; break; }
            const unsigned n = br.p[0] | (br.p[1] << 8), nn = br.p[2] | (br.p[3] << 8);
            br.p += 4;
            if ((n ^ nn) != 0xFFFFu || static_cast<int64_t>(n) > br.end - br.p) { err = "PNG: bad stored block"; return false; }
            out.insert(out.end(), br.p, br.p + n);
            br.p += n;
            continue;
        }
        if (type == 3) { err = "PNG: bad deflate block type"; return false; }
        uint8_t lengths[320];
        if (type == 1) {
            int i = 0;
            for (; i < 144; i++) lengths[i] = 8;
            for (; i < 256; i++) lengths[i] = 9;
            for (; i < 280; i++) lengths[i] = 7;
            for (; i < 288; i++) lengths[i] = 8;
            lit.build(lengths, 288);
            for (i = 0; i < 30; i++) lengths[i] = 5;
            dist.build(lengths, 30);
        } else {
            const int nlen = br.get(5) + 257, ndist = br.get(5) + 1, ncode = br.get(4) + 4;
            if (nlen > 286 || ndist > 30) { err = "PNG: bad deflate counts"; return false; }
            uint8_t cl[19] = {};
            for (int i = 0; i < ncode; i++) cl[order[i]] = static_cast<uint8_t>(br.get(3));
            Huffman clh;
            if (!clh.build(cl, 19)) { err = "PNG: bad code-length code"; return false; }
            int i = 0;
            while (i < nlen + ndist) {
                const int sym = br.decode(clh);
                if (sym < 0) { err = "PNG: bad code-length symbol"; return false; }
                if (sym < 16) {
                    lengths[i++] = static_cast<uint8_t>(sym);
                } else {
                    int prev = 0, rep;
                    if (sym == 16) {
                        if (i == 0) { err = "PNG: repeat with no previous length"; return false; }
                        prev = lengths[i - 1];
                        rep = 3 + br.get(2);
                    } else if (sym == 17) {
                        rep = 3 + br.get(3);
                    } else {
                        rep = 11 + br.get(7);
                    }
                    if (i + rep > nlen + ndist) { err = "PNG: length repeat overruns"; return false; }
                    while (rep--) lengths[i++] = static_cast<uint8_t>(prev);
                }
            }
            if (!br.ok) break;
            if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) { err = "PNG: bad Huffman table"; return false; }
        }
        for (;;) {
            const int sym = br.decode(lit);
            if (sym < 0 || !br.ok) { err = "PNG: bad literal/length symbol"; return false; }
            if (sym < 256) {
                out.push_back(static_cast<uint8_t>(sym));
            } else if (sym == 256) {
                break;
            } else {
                const int s = sym - 257;
                if (s >= 29) { err = "PNG: bad length symbol"; return false; }
                const int n = len_base[s] + br.get(len_extra[s]);
                const int ds = br.decode(dist);
                if (ds < 0 || ds >= 30) { err = "PNG: bad distance symbol"; return false; }
                const size_t d = dist_base[ds] + static_cast<size_t>(br.get(dist_extra[ds]));
                if (!br.ok || d > out.size()) { err = "PNG: distance too far back"; return false; }
                size_t from = out.size() - d;
                for (int k = 0; k < n; k++) out.push_back(out[from++]);
            }
        }
    } while (!last);
    if (!br.ok) { err = "PNG: truncated deflate stream"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------
// PNG
// ------------------------------------------------------------------------------------------
int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

struct PngInfo {
    int w = 0, h = 0, depth = 0, color = 0, interlace = 0;
    uint8_t palette[256][4];
    int pal_n = 0;
    bool has_key = false;
    uint16_t key[3] = {0, 0, 0};
    int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
};

// Unfilters one (sub-)image of pw x ph pixels starting at `data`, and writes its pixels into the
// full image at (x0 + i*dx, y0 + j*dy).  Returns the bytes consumed or -1.
int64_t png_pass(const PngInfo &pi, const uint8_t *data, int64_t avail, int pw, int ph, int x0, int y0, int dx, int dy,
                 Image &img) {
    if (pw == 0 || ph == 0) return 0;
    const int ch = pi.channels();
    const int bits_pp = ch * pi.depth;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter unit
    const int64_t stride = (static_cast<int64_t>(pw) * bits_pp + 7) / 8;
    if ((stride + 1) * ph > avail) return -1;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    const int maxv = (1 << pi.depth) - 1;
    for (int j = 0; j < ph; j++) {
        const uint8_t *row = data + (stride + 1) * j;
        const int ft = row[0];
        if (ft > 4) return -1;
        for (int64_t k = 0; k < stride; k++) {
            const int a = k >= bpp ? cur[k - bpp] : 0, b = prev[k], c = k >= bpp ? prev[k - bpp] : 0;
            int v = row[1 + k];
            switch (ft) {
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: break;
            }
            cur[k] = static_cast<uint8_t>(v);
        }
        const int y = y0 + j * dy;
        for (int i = 0; i < pw; i++) {
            uint16_t s[4] = {0, 0, 0, 0};  // raw samples
            if (pi.depth == 8) {
                for (int c = 0; c < ch; c++) s[c] = cur[static_cast<size_t>(i) * ch + c];
            } else if (pi.depth == 16) {
                for (int c = 0; c < ch; c++)
                    s[c] = static_cast<uint16_t>((cur[(static_cast<size_t>(i) * ch + c) * 2] << 8) |
                                                 cur[(static_cast<size_t>(i) * ch + c) * 2 + 1]);
            } else {
                const int64_t bit = static_cast<int64_t>(i) * pi.depth;
                s[0] = static_cast<uint16_t>((cur[bit >> 3] >> (8 - pi.depth - (bit & 7))) & maxv);
            }
            uint8_t *dst = &img.rgba[(static_cast<size_t>(y) * pi.w + (x0 + i * dx)) * 4];
            auto to8 = [&](uint16_t v) -> uint8_t {
                if (pi.depth == 8) return static_cast<uint8_t>(v);
                if (pi.depth == 16) return static_cast<uint8_t>(v >> 8);
                return static_cast<uint8_t>(v * 255 / maxv);
            };
            switch (pi.color) {
                case 0:
                    dst[0] = dst[1] = dst[2] = to8(s[0]);
                    dst[3] = (pi.has_key && s[0] == pi.key[0]) ? 0 : 255;
                    break;
                case 2:
                    dst[0] = to8(s[0]); dst[1] = to8(s[1]); dst[2] = to8(s[2]);
                    dst[3] = (pi.has_key && s[0] == pi.key[0] && s[1] == pi.key[1] && s[2] == pi.key[2]) ? 0 : 255;
                    break;
                case 3:
                    if (s[0] >= pi.pal_n) return -1;
                    std::memcpy(dst, pi.palette[s[0]], 4);
                    break;
                case 4:
                    dst[0] = dst[1] = dst[2] = to8(s[0]);
                    dst[3] = to8(s[1]);
                    break;
                default:
                    dst[0] = to8(s[0]); dst[1] = to8(s[1]); dst[2] = to8(s[2]); dst[3] = to8(s[3]);
                    break;
            }
        }
        prev.swap(cur);
    }
    return (stride + 1) * ph;
}

bool decode_png(const uint8_t *bytes, int64_t len, Image &img, std::string &err) {
    Reader r(bytes, len);
    r.skip(8);
    PngInfo pi;
    std::vector<uint8_t> idat;
    bool seen_ihdr = false, done = false;
    while (!done && r.ok && r.left() >= 8) {
        const uint32_t n = r.be32();
        const uint32_t type = r.be32();
        if (static_cast<int64_t>(n) + 4 > r.left()) { err = "PNG: truncated chunk"; return false; }
        const uint8_t *d = r.p;
        r.skip(static_cast<int64_t>(n) + 4);  // data + CRC (not verified: the inflate and size checks catch damage)
        switch (type) {
            case 0x49484452: {  // IHDR
                if (n != 13) { err = "PNG: bad IHDR"; return false; }
                Reader h(d, 13);
                pi.w = static_cast<int>(h.be32());
                pi.h = static_cast<int>(h.be32());
                pi.depth = h.u8();
                pi.color = h.u8();
                const int comp = h.u8(), filt = h.u8();
                pi.interlace = h.u8();
                const bool depth_ok = (pi.color == 0 && (pi.depth == 1 || pi.depth == 2 || pi.depth == 4 || pi.depth == 8 || pi.depth == 16)) ||
                                      (pi.color == 3 && (pi.depth == 1 || pi.depth == 2 || pi.depth == 4 || pi.depth == 8)) ||
                                      ((pi.color == 2 || pi.color == 4 || pi.color == 6) && (pi.depth == 8 || pi.depth == 16));
                if (pi.w < 1 || pi.h < 1 || pi.w > (1 << 24) || pi.h > (1 << 24) || !depth_ok || comp != 0 || filt != 0 ||
                    pi.interlace > 1) {
                    err = "PNG: unsupported IHDR";
                    return false;
                }
                seen_ihdr = true;
                break;
            }
            case 0x504C5445:  // PLTE
                if (n % 3 != 0 || n > 768) { err = "PNG: bad PLTE"; return false; }
                pi.pal_n = static_cast<int>(n / 3);
                for (int i = 0; i < pi.pal_n; i++) {
                    pi.palette[i][0] = d[i * 3];
                    pi.palette[i][1] = d[i * 3 + 1];
                    pi.palette[i][2] = d[i * 3 + 2];
                    pi.palette[i][3] = 255;
                }
                break;
            case 0x74524E53:  // tRNS
                if (!seen_ihdr) { err = "PNG: tRNS before IHDR"; return false; }
                if (pi.color == 3) {
                    for (uint32_t i = 0; i < n && i < 256; i++) pi.palette[i][3] = d[i];
                } else if (pi.color == 0 && n >= 2) {
                    pi.has_key = true;
                    pi.key[0] = static_cast<uint16_t>((d[0] << 8) | d[1]);
                } else if (pi.color == 2 && n >= 6) {
                    pi.has_key = true;
                    for (int c = 0; c < 3; c++) pi.key[c] = static_cast<uint16_t>((d[c * 2] << 8) | d[c * 2 + 1]);
                }
                break;
            case 0x49444154:  // IDAT
                idat.insert(idat.end(), d, d + n);
                break;
            case 0x49454E44:  // IEND
                done = true;
                break;
            default:
                break;  // ancillary chunks
        }
    }
    if (!seen_ihdr || idat.empty()) { err = "PNG: missing IHDR or IDAT"; return false; }
    if (pi.color == 3 && pi.pal_n == 0) { err = "PNG: paletted image without PLTE"; return false; }
    if (static_cast<int64_t>(pi.w) * pi.h > (int64_t{1} << 30)) { err = "PNG: image too large"; return false; }

    std::vector<uint8_t> raw;
    const size_t expect = (static_cast<size_t>(pi.w) * pi.channels() * pi.depth / 8 + 2) * pi.h;
    if (!inflate_zlib(idat.data(), static_cast<int64_t>(idat.size()), raw, expect, err)) return false;

    img.w = pi.w;
    img.h = pi.h;
    img.rgba.assign(static_cast<size_t>(pi.w) * pi.h * 4, 0);
    if (!pi.interlace) {
        if (png_pass(pi, raw.data(), static_cast<int64_t>(raw.size()), pi.w, pi.h, 0, 0, 1, 1, img) < 0) {
            err = "PNG: bad scanline data";
            return false;
        }
    } else {
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1};
        static const int dxs[7] = {8, 8, 4, 4, 2, 2, 1}, dys[7] = {8, 8, 8, 4, 4, 2, 2};
        int64_t off = 0;
        for (int p = 0; p < 7; p++) {
            const int pw = (pi.w - xs[p] + dxs[p] - 1) / dxs[p], ph = (pi.h - ys[p] + dys[p] - 1) / dys[p];
            const int64_t used = png_pass(pi, raw.data() + off, static_cast<int64_t>(raw.size()) - off, pw > 0 ? pw : 0,
                                          ph > 0 ? ph : 0, xs[p], ys[p], dxs[p], dys[p], img);
            if (used < 0) { err = "PNG: bad interlaced data"; return false; }
            off += used;
        }
    }
    return true;
}

// ------------------------------------------------------------------------------------------
// writers
// ------------------------------------------------------------------------------------------
uint32_t crc32(uint32_t crc, const uint8_t *d, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 255] ^ (crc >> 8);
    return ~crc;
}

bool write_all(const char *path, const std::vector<uint8_t> &bytes, std::string &err) {
    FILE *f = std::fopen(path, "wb");
    if (!f) { err = std::string("cannot open for writing: ") + path; return false; }
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    if (std::fclose(f) != 0 || !ok) { err = std::string("short write: ") + path; return false; }
    return true;
}

}  // namespace

void Image::flip_rows() {
    const size_t stride = static_cast<size_t>(w) * 4;
    std::vector<uint8_t> tmp(stride);
    for (int j = 0; j < h / 2; j++) {
        uint8_t *a = &rgba[stride * j], *b = &rgba[stride * (h - 1 - j)];
        std::memcpy(tmp.data(), a, stride);
        std::memcpy(a, b, stride);
        std::memcpy(b, tmp.data(), stride);
    }
}

bool Image::decode(const uint8_t *bytes, int64_t len, bool invert_y, std::string &err) {
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    bool ok;
    if (len >= 8 && std::memcmp(bytes, png_sig, 8) == 0) {
        ok = decode_png(bytes, len, *this, err);
    } else if (len >= 18) {
        ok = decode_tga(bytes, len, *this, err);  // TGA has no signature: it is the fallback
    } else {
        err = "not a PNG or TGA image";
        ok = false;
    }
    if (!ok) {
        w = h = 0;
        rgba.clear();
        return false;
    }
    if (invert_y) flip_rows();
    return true;
}

bool Image::load(const std::string &path, bool invert_y, std::string &err) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { err = "Can't open file: " + path; return false; }
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return decode(bytes.data(), static_cast<int64_t>(bytes.size()), invert_y, err);
}

bool write_tga(const char *path, const uint8_t *rgba_bottom_up, int w, int h, bool rle, std::string &err) {
    if (!rgba_bottom_up || w < 1 || h < 1 || w > 65535 || h > 65535) { err = "write_tga: bad arguments"; return false; }
    std::vector<uint8_t> out;
    out.reserve(18 + static_cast<size_t>(w) * h * 4);
    const uint8_t head[18] = {0, 0, static_cast<uint8_t>(rle ? 10 : 2), 0, 0, 0, 0, 0, 0, 0, 0, 0,
                              static_cast<uint8_t>(w & 255), static_cast<uint8_t>(w >> 8),
                              static_cast<uint8_t>(h & 255), static_cast<uint8_t>(h >> 8), 32, 8};
    out.insert(out.end(), head, head + 18);
    auto put = [&](const uint8_t *p) {
        out.push_back(p[2]); out.push_back(p[1]); out.push_back(p[0]); out.push_back(p[3]);
    };
    for (int y = 0; y < h; y++) {
        const uint8_t *row = rgba_bottom_up + static_cast<size_t>(y) * w * 4;
        if (!rle) {
            for (int x = 0; x < w; x++) put(row + x * 4);
            continue;
        }
        int x = 0;
        while (x < w) {  // packets never cross a row
            int run = 1;
            while (x + run < w && run < 128 && std::memcmp(row + x * 4, row + (x + run) * 4, 4) == 0) run++;
            if (run >= 2) {
                out.push_back(static_cast<uint8_t>(128 | (run - 1)));
                put(row + x * 4);
                x += run;
            } else {
                int n = 1;
                while (x + n < w && n < 128 &&
                       !(x + n + 1 < w && std::memcmp(row + (x + n) * 4, row + (x + n + 1) * 4, 4) == 0))
                    n++;
                out.push_back(static_cast<uint8_t>(n - 1));
                for (int k = 0; k < n; k++) put(row + (x + k) * 4);
                x += n;
            }
        }
    }
    return write_all(path, out, err);
}

bool write_png(const char *path, const uint8_t *rgba_bottom_up, int w, int h, std::string &err) {
    if (!rgba_bottom_up || w < 1 || h < 1) { err = "write_png: bad arguments"; return false; }
    // Scanlines top-down, filter 0, in stored (uncompressed) deflate blocks.
    const size_t stride = static_cast<size_t>(w) * 4 + 1;
    std::vector<uint8_t> raw(stride * h);
    for (int y = 0; y < h; y++) {
        raw[stride * y] = 0;
        std::memcpy(&raw[stride * y + 1], rgba_bottom_up + static_cast<size_t>(h - 1 - y) * w * 4, stride - 1);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78);
    z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back(static_cast<uint8_t>(n & 255));
        z.push_back(static_cast<uint8_t>(n >> 8));
        z.push_back(static_cast<uint8_t>(~n & 255));
        z.push_back(static_cast<uint8_t>((~n >> 8) & 255));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) {
            a = (a + raw[off + i]) % 65521u;
            b = (b + a) % 65521u;
        }
        off += n;
    }
    const uint32_t adler = (b << 16) | a;
    for (int s = 24; s >= 0; s -= 8) z.push_back(static_cast<uint8_t>(adler >> s));

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char *type, const std::vector<uint8_t> &data) {
        const uint32_t n = static_cast<uint32_t>(data.size());
        for (int s = 24; s >= 0; s -= 8) out.push_back(static_cast<uint8_t>(n >> s));
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        const uint32_t c = crc32(0, &out[start], out.size() - start);
        for (int s = 24; s >= 0; s -= 8) out.push_back(static_cast<uint8_t>(c >> s));
    };
    std::vector<uint8_t> ihdr;
    for (int s = 24; s >= 0; s -= 8) ihdr.push_back(static_cast<uint8_t>(static_cast<uint32_t>(w) >> s));
    for (int s = 24; s >= 0; s -= 8) ihdr.push_back(static_cast<uint8_t>(static_cast<uint32_t>(h) >> s));
    const uint8_t tail[5] = {8, 6, 0, 0, 0};
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    return write_all(path, out, err);
}

}  // namespace rth
