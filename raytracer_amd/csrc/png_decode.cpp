// png_decode.cpp -- PNG -> RGBA8, standing in for Go's image/png + image/draw
// as used by NewTexture (models/texture.go:24-46).  Self-contained: zlib
// inflate (stored, fixed and dynamic Huffman blocks), the five PNG row filters,
// colour types 0/2/3/4/6 at 8 or 16 bits per channel and palette / grey images
// at 1, 2, 4, 8 bits, non-interlaced.
//
// The reference keeps only two decoder outputs: opaque truecolour (Go decodes
// it straight into image.RGBA) and truecolour with alpha (image.NRGBA, then
// image/draw converts to premultiplied image.RGBA); other colour models leave
// its texture pointer nil.  This decoder gives the same bytes for those two
// and, instead of failing, expands grey and palette images to the same form.
#include <cstring>

#include "backend.h"

namespace rt {
namespace {

struct BitReader {
    const uint8_t *p, *end;
    uint32_t bits = 0;
    int n = 0;
    bool ok = true;
    uint32_t get(int count) {
        while (n < count) {
            if (p >= end) {
                ok = false;
                return 0;
            }
            bits |= (uint32_t)(*p++) << n;
            n += 8;
        }
        const uint32_t v = bits & ((count == 32) ? 0xFFFFFFFFu : ((1u << count) - 1));
        bits >>= count;
        n -= count;
        return v;
    }
    void align() {
        bits = 0;
        n = 0;
    }
};

struct Huffman {
    uint16_t count[16];
    uint16_t symbol[320];
    bool build(const uint8_t *lengths, int n) {
        std::memset(count, 0, sizeof count);
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = (uint16_t)i;
        return true;
    }
    int decode(BitReader &br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= (int)br.get(1);
            if (!br.ok) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

// Inflates until the stream ends or `limit` bytes are out (what the image
// needs; a stream that expands further is not followed).
bool inflate(const uint8_t *data, size_t size, std::vector<uint8_t> &out, size_t limit) {
    if (size < 2) return false;
    BitReader br{data + 2, data + size};   // skip the zlib header
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
        const uint32_t last = br.get(1), type = br.get(2);
        if (!br.ok) return false;
        if (type == 0) {
            br.align();
            if (br.end - br.p < 4) return false;
            const uint32_t len = br.p[0] | (br.p[1] << 8);
            br.p += 4;
            if ((size_t)(br.end - br.p) < len) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; ++i) lengths[i] = 8;
                for (; i < 256; ++i) lengths[i] = 9;
                for (; i < 280; ++i) lengths[i] = 7;
                for (; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int nlen = (int)br.get(5) + 257, ndist = (int)br.get(5) + 1, ncode = (int)br.get(4) + 4;
                if (!br.ok || nlen > 286 || ndist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = (uint8_t)br.get(3);
                Huffman lencode;
                if (!lencode.build(cl, 19)) return false;
                int idx = 0;
                while (idx < nlen + ndist) {
                    const int sym = lencode.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lengths[idx++] = (uint8_t)sym;
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) return false;
                            val = lengths[idx - 1];
                            rep = 3 + (int)br.get(2);
                        } else if (sym == 17) {
                            rep = 3 + (int)br.get(3);
                        } else {
                            rep = 11 + (int)br.get(7);
                        }
                        if (idx + rep > nlen + ndist) return false;
                        while (rep--) lengths[idx++] = (uint8_t)val;
                    }
                }
                if (!lit.build(lengths, nlen)) return false;
                if (!dist.build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (out.size() >= limit) break;
                if (sym < 256) {
                    out.push_back((uint8_t)sym);
                } else if (sym == 256) {
                    break;
                } else {
                    const int li = sym - 257;
                    if (li >= 29) return false;
                    const int len = lbase[li] + (int)br.get(lext[li]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dbase[ds] + br.get(dext[ds]);
                    if (!br.ok || d > out.size()) return false;
                    const size_t start = out.size() - d;
                    for (int i = 0; i < len; ++i) out.push_back(out[start + (size_t)i]);
                }
            }
        } else {
            return false;
        }
        if (last || out.size() >= limit) break;
    }
    return true;
}

inline uint32_t be32(const uint8_t *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

}  // namespace

rt_status decode_png(const uint8_t *data, size_t size, std::vector<uint8_t> &rgba, int32_t &width,
                     int32_t &height) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (!data || size < 33 || std::memcmp(data, sig, 8) != 0) return fail(RT_ERR_PARSE, "not a PNG file");
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool seen_end = false;
    while (pos + 12 <= size && !seen_end) {
        const uint32_t len = be32(data + pos);
        const uint8_t *type = data + pos + 4;
        if (pos + 12 + (size_t)len > size) return fail(RT_ERR_PARSE, "truncated PNG chunk");
        const uint8_t *body = data + pos + 8;
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return fail(RT_ERR_PARSE, "bad PNG header");
            w = be32(body);
            h = be32(body + 4);
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || w > 65535 || h > 65535) return fail(RT_ERR_PARSE, "unsupported PNG size");
    if (interlace != 0) return fail(RT_ERR_PARSE, "interlaced PNG is not supported");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return fail(RT_ERR_PARSE, "unsupported PNG colour type");
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return fail(RT_ERR_PARSE, "unsupported PNG bit depth");
    if (ctype == 3 && depth == 16) return fail(RT_ERR_PARSE, "unsupported PNG bit depth");
    const size_t bpp_bits = (size_t)channels * (size_t)depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;
    std::vector<uint8_t> raw;
    const size_t need = (stride + 1) * (size_t)h;
    // deflate expands at most 1032-fold: a short stream cannot fill a large image
    if (idat.size() < need / 1032) return fail(RT_ERR_PARSE, "corrupt PNG data stream");
    raw.reserve(need + 258);
    if (!inflate(idat.data(), idat.size(), raw, need) || raw.size() < need)
        return fail(RT_ERR_PARSE, "corrupt PNG data stream");
    // undo the row filters in place
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *row = &raw[(stride + 1) * (size_t)y];
        const int filter = row[0];
        uint8_t *cur = row + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
            int v = cur[i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: return fail(RT_ERR_PARSE, "bad PNG row filter");
            }
            cur[i] = (uint8_t)v;
        }
        std::memcpy(prev.data(), cur, stride);
    }
    width = (int32_t)w;
    height = (int32_t)h;
    rgba.assign((size_t)w * h * 4, 255);
    const int step = depth == 16 ? 2 : 1;   // 16-bit samples: keep the high byte
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t *cur = &raw[(stride + 1) * (size_t)y + 1];
        for (uint32_t x = 0; x < w; ++x) {
            uint8_t px[4] = {0, 0, 0, 255};
            if (ctype == 2 || ctype == 6) {
                const uint8_t *s = cur + (size_t)x * (size_t)channels * (size_t)step;
                px[0] = s[0];
                px[1] = s[step];
                px[2] = s[2 * step];
                if (ctype == 6) px[3] = s[3 * step];
            } else {
                uint32_t v;
                if (depth >= 8) {
                    v = cur[(size_t)x * (size_t)channels * (size_t)step];
                } else {
                    const size_t bit = (size_t)x * (size_t)depth;
                    v = (cur[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1u << depth) - 1);
                }
                if (ctype == 3) {
                    if ((size_t)v * 3 + 2 < palette.size()) {
                        px[0] = palette[v * 3];
                        px[1] = palette[v * 3 + 1];
                        px[2] = palette[v * 3 + 2];
                    }
                    if (v < trns.size()) px[3] = trns[v];
                } else {
                    const uint8_t g = depth < 8 ? (uint8_t)(v * 255 / ((1u << depth) - 1)) : (uint8_t)v;
                    px[0] = px[1] = px[2] = g;
                    if (ctype == 4) px[3] = cur[((size_t)x * 2 + 1) * (size_t)step];
                }
            }
            uint8_t *d = &rgba[((size_t)y * w + x) * 4];
            if (px[3] != 255) {
                // image/draw NRGBA -> RGBA: premultiply in 16-bit space
                const uint32_t sa = (uint32_t)px[3] * 0x101;
                d[0] = (uint8_t)(((uint32_t)px[0] * sa / 0xff) >> 8);
                d[1] = (uint8_t)(((uint32_t)px[1] * sa / 0xff) >> 8);
                d[2] = (uint8_t)(((uint32_t)px[2] * sa / 0xff) >> 8);
                d[3] = px[3];
            } else {
                d[0] = px[0];
                d[1] = px[1];
                d[2] = px[2];
                d[3] = 255;
            }
        }
    }
    return RT_OK;
}

}  // namespace rt
