// image_decode.cc — image bytes -> 8-bit RGB for the LLaVA path (clip.h clip_image_load_from_bytes; the reference calls stb_image through
// clip_image_load_from_bytes, /root/reference/src/llama_server_context.cc:568).  Own decoders, written from the format specifications:
//   PNG  (RFC 2083 + zlib / deflate RFC 1950 / 1951): 8-bit grey, grey + alpha, RGB, RGBA, palette (1 / 2 / 4 / 8 bit); 16-bit samples keep their high byte;
//        non-interlaced or Adam7; alpha is dropped (stb_image's 3-channel request does the same)
//   BMP  uncompressed 24 / 32 bit, bottom-up or top-down
//   PNM  binary P5 / P6, maxval < 256
//   JPEG Huffman-coded sequential (SOF0 / SOF1) and progressive (SOF2) DCT, 8-bit, 1 or 3 components with any sampling factors, any number of scans
//        (interleaved or not), restart intervals; lossless / hierarchical / arithmetic-coded files are refused
// Lossless formats decode to the bytes any decoder produces.  JPEG: the inverse DCT here is the float reference transform rounded to nearest, 2:1 subsampled
// chroma goes through libjpeg's triangle filter; stb_image uses a fixed-point transform and its own variant of that filter, so its bytes can differ from these by
// a few units (as two conforming JPEG decoders do).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "clip.h"
#include "parallel_rows.h"

namespace mi355 {
namespace {

// the bytes come from a request: a header may not make the server allocate gigabytes (48 M pixels = an 8000 x 6000 photograph; the tower sees 336 x 336 of it)
constexpr int64_t MAX_PIXELS = 48ll << 20;

// ------------------------------------------------------------------------------------------------ inflate
struct BitReader {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int cnt = 0;
    bool bad = false;
    int bits(int n) {
        while (cnt < n) {
            if (p >= end) { bad = true; return 0; }
            buf |= (uint32_t)(*p++) << cnt;
            cnt += 8;
        }
        const int v = (int)(buf & ((1u << n) - 1u));
        buf >>= n; cnt -= n;
        return v;
    }
};
struct Huff {
    uint16_t count[16] = {0}, symbol[288] = {0};
    uint16_t fast[512];                    // the next 9 bits of the stream -> (length << 9 | symbol) of a code of at most 9 bits; 0 = a longer code, or none
    void build(const uint8_t *len, int n) {
        memset(count, 0, sizeof count);
        memset(fast, 0, sizeof fast);
        for (int i = 0; i < n; i++) count[len[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        uint32_t next[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = (uint16_t)(offs[i] + count[i]);
        uint32_t code = 0;
        for (int l = 1; l <= 15; l++) { next[l] = code; code = (code + count[l]) << 1; }
        for (int i = 0; i < n; i++) {
            const int l = len[i];
            if (!l) continue;
            symbol[offs[l]++] = (uint16_t)i;
            const uint32_t c = next[l]++;
            if (l > 9 || c >= (1u << l)) continue;
            uint32_t rev = 0;                                   // (codes are packed starting from their most significant bit, the stream is read from bit 0)
            for (int b = 0; b < l; b++) rev |= ((c >> b) & 1u) << (l - 1 - b);
            for (uint32_t j = rev; j < 512; j += 1u << l) fast[j] = (uint16_t)(l << 9 | i);
        }
    }
    int decode(BitReader &br) const {
        while (br.cnt < 9 && br.p < br.end) { br.buf |= (uint32_t)(*br.p++) << br.cnt; br.cnt += 8; }
        const uint16_t f = fast[br.buf & 511u];
        if (f && (f >> 9) <= br.cnt) { const int l = f >> 9; br.buf >>= l; br.cnt -= l; return f & 511; }
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= 15; l++) {
            code |= br.bits(1);
            if (br.bad) return -1;
            const int c = count[l];
            if (code - c < first) return symbol[index + (code - first)];
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};
bool inflate_zlib(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t expect) {
    if (n < 6) return false;
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8 | src[1]) % 31) != 0 || (src[1] & 0x20)) return false;
    BitReader br{src + 2, src + n};
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    out.clear();
    out.reserve(expect);
    int last;
    do {
        last = br.bits(1);
        const int type = br.bits(2);
        if (br.bad) return false;
        if (type == 0) {
            // a stored block starts at the next byte boundary of the STREAM: the Huffman decoder of a preceding block looks up to 16 bits ahead, so whole
            // bytes (LEN / NLEN) may already sit in the bit buffer - hand them back before dropping the partial byte
            br.p -= (br.cnt >> 3);
            br.buf = 0; br.cnt = 0;
            if (br.p + 4 > br.end) return false;
            const unsigned len = br.p[0] | br.p[1] << 8, nlen = br.p[2] | br.p[3] << 8;
            br.p += 4;
            if ((len ^ 0xffffu) != nlen || br.p + len > br.end) return false;
            out.insert(out.end(), br.p, br.p + len);
            br.p += len;
        } else if (type == 1 || type == 2) {
            Huff hl, hd;
            uint8_t lens[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lens[i] = 8;
                for (; i < 256; i++) lens[i] = 9;
                for (; i < 280; i++) lens[i] = 7;
                for (; i < 288; i++) lens[i] = 8;
                hl.build(lens, 288);
                for (i = 0; i < 30; i++) lens[i] = 5;
                hd.build(lens, 30);
            } else {
                const int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
                if (br.bad || nlen > 286 || ndist > 30) return false;
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
                Huff hc;
                hc.build(cl, 19);
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = hc.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) lens[i++] = (uint8_t)sym;
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + br.bits(2); }
                        else if (sym == 17) rep = 3 + br.bits(3);
                        else rep = 11 + br.bits(7);
                        if (br.bad || i + rep > nlen + ndist) return false;
                        while (rep--) lens[i++] = (uint8_t)val;
                    }
                }
                if (lens[256] == 0) return false;
                hl.build(lens, nlen);
                hd.build(lens + nlen, ndist);
            }
            for (;;) {
                const int sym = hl.decode(br);
                if (sym < 0) return false;
                if (sym < 256) out.push_back((uint8_t)sym);
                else if (sym == 256) break;
                else {
                    if (sym > 285) return false;
                    const int len = lbase[sym - 257] + br.bits(lext[sym - 257]);
                    const int ds = hd.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const size_t dist = (size_t)dbase[ds] + (size_t)br.bits(dext[ds]);
                    if (br.bad || dist > out.size()) return false;
                    const size_t from = out.size() - dist;
                    for (int k = 0; k < len; k++) out.push_back(out[from + (size_t)k]);
                }
                if (out.size() > expect + 65536) return false;       // (a bomb, or not this image's stream)
            }
        } else return false;
    } while (!last);
    return true;
}

uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// ------------------------------------------------------------------------------------------------ PNG
std::string load_png(const uint8_t *d, size_t n, ClipImageU8 &out) {
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, pal;
    bool have_ihdr = false;
    while (pos + 12 <= n) {
        const uint32_t len = be32(d + pos);
        const uint8_t *tag = d + pos + 4, *body = d + pos + 8;
        if (len > n || pos + 12 + (size_t)len > n) return "PNG: truncated chunk";
        if (!memcmp(tag, "IHDR", 4)) {
            if (len != 13) return "PNG: bad header";
            w = (int)be32(body); h = (int)be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) return "PNG: unknown compression / filter method";
            have_ihdr = true;
        } else if (!memcmp(tag, "PLTE", 4)) pal.assign(body, body + len);
        else if (!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(tag, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0 || w > 16384 || h > 16384 || (int64_t)w * h > MAX_PIXELS) return "PNG: bad dimensions";
    if (interlace > 1) return "PNG: unknown interlace method";
    int ch;
    switch (ctype) {
        case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break;
        default: return "PNG: bad colour type";
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) return "PNG: unsupported bit depth";
    if (ctype == 3 && (depth == 16 || pal.empty())) return "PNG: palette image without a palette";
    const int bpp_bits = ch * depth, bpp = (bpp_bits + 7) / 8;
    // the picture is one pass of w x h pixels, or (Adam7) seven reduced pictures - pass p holds the pixels (x0 + i * dx, y0 + j * dy) - each filtered on its own
    struct Pass { int x0, y0, dx, dy, pw, ph; size_t stride; };
    static const int A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    std::vector<Pass> passes;
    size_t need = 0;
    for (int p = 0; p < (interlace ? 7 : 1); p++) {
        Pass ps;
        if (interlace) { ps.x0 = A7[p][0]; ps.y0 = A7[p][1]; ps.dx = A7[p][2]; ps.dy = A7[p][3]; }
        else { ps.x0 = ps.y0 = 0; ps.dx = ps.dy = 1; }
        ps.pw = (w - ps.x0 + ps.dx - 1) / ps.dx; ps.ph = (h - ps.y0 + ps.dy - 1) / ps.dy;
        if (ps.pw <= 0 || ps.ph <= 0) continue;
        ps.stride = ((size_t)ps.pw * bpp_bits + 7) / 8;
        need += (ps.stride + 1) * (size_t)ps.ph;
        passes.push_back(ps);
    }
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat.data(), idat.size(), raw, need) || raw.size() < need) return "PNG: bad compressed data";
    out.nx = w; out.ny = h; out.rgb.assign((size_t)3 * w * h, 0);
    const uint8_t *line = raw.data();
    for (const Pass &ps : passes) {
        const size_t stride = ps.stride;
        std::vector<uint8_t> prev(stride, 0), cur(stride);
        for (int y = 0; y < ps.ph; y++, line += stride + 1) {
            const int ft = line[0];
            if (ft > 4) return "PNG: bad filter type";
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
                int pred = 0;
                switch (ft) {
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) >> 1; break;
                    case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                    default: break;
                }
                cur[i] = (uint8_t)(line[1 + i] + pred);
            }
            uint8_t *orow = out.rgb.data() + (size_t)3 * (size_t)(ps.y0 + y * ps.dy) * w;
            for (int x = 0; x < ps.pw; x++) {
                auto sample = [&](int k) -> int {                     // sample k of pixel x of this pass, reduced to 8 bits
                    if (depth == 8) return cur[(size_t)x * ch + k];
                    if (depth == 16) return cur[((size_t)x * ch + k) * 2];
                    const int bit = x * depth, v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
                    return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
                };
                uint8_t *o = orow + (size_t)3 * (size_t)(ps.x0 + x * ps.dx);
                if (ctype == 3) {
                    const size_t idx = (size_t)sample(0);
                    if (idx * 3 + 2 >= pal.size()) return "PNG: palette index out of range";
                    o[0] = pal[idx * 3]; o[1] = pal[idx * 3 + 1]; o[2] = pal[idx * 3 + 2];
                } else if (ch <= 2) { const uint8_t g = (uint8_t)sample(0); o[0] = g; o[1] = g; o[2] = g; }
                else { o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2); }
            }
            prev.swap(cur);
        }
    }
    return "";
}

// ------------------------------------------------------------------------------------------------ BMP / PNM
std::string load_bmp(const uint8_t *d, size_t n, ClipImageU8 &out) {
    if (n < 54) return "BMP: truncated";
    auto le32 = [&](size_t o) { return (uint32_t)d[o] | (uint32_t)d[o + 1] << 8 | (uint32_t)d[o + 2] << 16 | (uint32_t)d[o + 3] << 24; };
    const uint32_t off = le32(10), hsz = le32(14);
    const int w = (int)le32(18);
    int h = (int)le32(22);
    const int bpp = d[28] | d[29] << 8;
    const uint32_t comp = le32(30);
    if (hsz < 40 || (bpp != 24 && bpp != 32) || (comp != 0 && comp != 3) || w <= 0 || h == 0 || w > 16384) return "BMP: only uncompressed 24 / 32-bit images are supported";
    const bool top_down = h < 0;
    if (top_down) h = -h;
    if (h > 16384 || (int64_t)w * h > MAX_PIXELS) return "BMP: bad dimensions";
    const size_t stride = (((size_t)w * bpp / 8) + 3) & ~(size_t)3;
    if ((size_t)off + stride * (size_t)h > n) return "BMP: truncated pixel data";
    out.nx = w; out.ny = h; out.rgb.resize((size_t)3 * w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *line = d + off + stride * (size_t)(top_down ? y : h - 1 - y);
        for (int x = 0; x < w; x++) {
            const uint8_t *px = line + (size_t)x * (bpp / 8);
            uint8_t *o = out.rgb.data() + 3 * ((size_t)y * w + x);
            o[0] = px[2]; o[1] = px[1]; o[2] = px[0];
        }
    }
    return "";
}
std::string load_pnm(const uint8_t *d, size_t n, ClipImageU8 &out) {
    const int ch = d[1] == '6' ? 3 : 1;
    size_t pos = 2;
    int vals[3], got = 0;
    while (got < 3 && pos < n) {
        while (pos < n && (d[pos] == ' ' || d[pos] == '\n' || d[pos] == '\r' || d[pos] == '\t')) pos++;
        if (pos < n && d[pos] == '#') { while (pos < n && d[pos] != '\n') pos++; continue; }
        int v = 0, digits = 0;
        while (pos < n && d[pos] >= '0' && d[pos] <= '9' && digits < 9) { v = v * 10 + (d[pos] - '0'); pos++; digits++; }
        if (!digits) return "PNM: bad header";
        vals[got++] = v;
    }
    if (got < 3 || pos >= n) return "PNM: bad header";
    pos++;                                                       // the single whitespace byte behind maxval
    const int w = vals[0], h = vals[1];
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || (int64_t)w * h > MAX_PIXELS || vals[2] <= 0 || vals[2] > 255) return "PNM: unsupported dimensions or maxval";
    if (pos + (size_t)w * h * ch > n) return "PNM: truncated pixel data";
    out.nx = w; out.ny = h; out.rgb.resize((size_t)3 * w * h);
    for (size_t i = 0; i < (size_t)w * h; i++)
        for (int k = 0; k < 3; k++) out.rgb[3 * i + k] = (uint8_t)((int)d[pos + i * ch + (ch == 3 ? k : 0)] * 255 / vals[2]);
    return "";
}

// ------------------------------------------------------------------------------------------------ JPEG
struct JHuff {
    uint8_t bits[17] = {0}, vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    uint16_t fast[512];                    // the next 9 bits -> (length << 8 | value) of a code of at most 9 bits, 0 = longer code (or none)
    void build() {
        int code = 0, k = 0;
        memset(fast, 0, sizeof fast);
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k; mincode[l] = code;
            if (l <= 9)
                for (int i = 0; i < bits[l] && code + i < (1 << l); i++) {
                    const int first = (code + i) << (9 - l);
                    for (int j = 0; j < (1 << (9 - l)); j++) fast[first + j] = (uint16_t)(l << 8 | vals[k + i]);
                }
            code += bits[l]; k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};
struct JComp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0, stride = 0, rows = 0, bw = 0, bh = 0;     // bw x bh blocks (whole MCUs)
    uint16_t q[64] = {0};
    bool q_latched = false;
    std::vector<int16_t> coef;             // [bh][bw][64], natural order, not yet dequantised
    std::vector<uint8_t> plane;
};
struct JBits {
    const uint8_t *p, *end;
    uint32_t buf = 0;
    int cnt = 0;
    bool bad = false;
    int marker = 0;
    void fill() {
        while (cnt <= 24) {
            int b = 0;
            if (marker == 0 && p < end) {
                b = *p++;
                if (b == 0xff) {
                    int m = p < end ? *p : 0;
                    if (m == 0) p++;
                    else { marker = m; p++; b = 0; }
                }
            }
            buf |= (uint32_t)b << (24 - cnt);
            cnt += 8;
        }
    }
    int get(int n) {
        if (n == 0) return 0;
        if (cnt < n) fill();
        const int v = (int)(buf >> (32 - n));
        buf <<= n; cnt -= n;
        return v;
    }
    void reset() { buf = 0; cnt = 0; marker = 0; }
};
int jdecode(JBits &br, const JHuff &h) {
    if (br.cnt < 9) br.fill();
    const uint16_t f = h.fast[br.buf >> 23];
    if (f) { const int l = f >> 8; br.buf <<= l; br.cnt -= l; return f & 255; }
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.get(1);
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    br.bad = true;
    return 0;
}
int jextend(int v, int t) { return t == 0 ? 0 : (v < (1 << (t - 1)) ? v - (1 << t) + 1 : v); }
inline uint8_t to_byte(float v) { return (uint8_t)(v < 0.0f ? 0 : v > 255.0f ? 255 : (int)(v + 0.5f)); }      // round to nearest, clamp
void idct8x8(const float *in, uint8_t *out, int stride) {          // the reference (separable, float) inverse transform, level shift, round, clamp
    static float c[8][8];
    static bool init = false;
    if (!init) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? (float)M_SQRT1_2 : 1.0f) * 0.5f * cosf((2 * x + 1) * u * (float)M_PI / 16.0f);
        init = true;
    }
    // most of a block's coefficients are zero (the high frequencies of both directions): the sums run over the occupied corner only, eight outputs of a row at
    // a time (the inner loops are over x: contiguous, vector width)
    static float ct[8][8];                 // ct[u][x] = c[x][u]
    static bool init2 = false;
    if (!init2) {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) ct[u][x] = c[x][u];
        init2 = true;
    }
    int vmax = -1, umax = 0;
    for (int v = 0; v < 8; v++)
        for (int u = 0; u < 8; u++)
            if (in[v * 8 + u] != 0.0f) { vmax = v; if (u > umax) umax = u; }
    if (vmax < 0) {
        for (int y = 0; y < 8; y++) memset(out + y * stride, 128, 8);
        return;
    }
    float tmp[64];
    for (int v = 0; v <= vmax; v++) {
        float acc[8] = {0};
        for (int u = 0; u <= umax; u++) { const float a = in[v * 8 + u]; for (int x = 0; x < 8; x++) acc[x] += ct[u][x] * a; }
        for (int x = 0; x < 8; x++) tmp[v * 8 + x] = acc[x];
    }
    for (int y = 0; y < 8; y++) {
        float acc[8] = {0};
        for (int v = 0; v <= vmax; v++) { const float a = c[y][v]; for (int x = 0; x < 8; x++) acc[x] += a * tmp[v * 8 + x]; }
        for (int x = 0; x < 8; x++) out[y * stride + x] = to_byte(acc[x] + 128.0f);
    }
}
// one block of a progressive scan (ITU T.81 annex G): spectral selection [Ss, Se], successive approximation (Ah = the bit position of the previous scan over these
// coefficients, 0 in a first scan; Al = this scan's).  coef in natural (row-major) order; eobrun = blocks still covered by an end-of-band run
struct JScan { int Ss = 0, Se = 63, Ah = 0, Al = 0, eobrun = 0; };
const uint8_t JZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                         35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const char *jblock_sequential(JBits &br, const JHuff &dc, const JHuff &ac, int &pred, int16_t *coef) {
    const int t = jdecode(br, dc);
    if (t > 11) return "JPEG: bad DC code";
    pred = (int)((unsigned)pred + (unsigned)jextend(br.get(t), t));
    coef[0] = (int16_t)pred;
    for (int k = 1; k < 64;) {
        const int rs = jdecode(br, ac), r = rs >> 4, sz = rs & 15;
        if (br.bad) return "JPEG: bad Huffman code";
        if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
        k += r;
        if (k > 63) return "JPEG: coefficient index out of range";
        coef[JZZ[k]] = (int16_t)jextend(br.get(sz), sz);
        k++;
    }
    return nullptr;
}
const char *jblock_progressive(JBits &br, const JHuff &dc, const JHuff &ac, int &pred, int16_t *coef, JScan &sc) {
    if (sc.Ss == 0) {                       // DC scan
        if (sc.Ah == 0) {
            const int t = jdecode(br, dc);
            if (t > 11) return "JPEG: bad DC code";
            pred = (int)((unsigned)pred + (unsigned)jextend(br.get(t), t));     // (hostile input: wraps instead of overflowing)
            coef[0] = (int16_t)((unsigned)pred << sc.Al);
        } else if (br.get(1)) coef[0] = (int16_t)(coef[0] | (1 << sc.Al));
        return nullptr;
    }
    if (sc.Ah == 0) {                       // AC, first scan over this band
        if (sc.eobrun > 0) { sc.eobrun--; return nullptr; }
        for (int k = sc.Ss; k <= sc.Se;) {
            const int rs = jdecode(br, ac), r = rs >> 4, sz = rs & 15;
            if (br.bad) return "JPEG: bad Huffman code";
            if (sz == 0) {
                if (r < 15) { sc.eobrun = (1 << r) - 1; if (r) sc.eobrun += br.get(r); break; }
                k += 16;
                continue;
            }
            k += r;
            if (k > sc.Se) return "JPEG: coefficient index out of range";
            coef[JZZ[k]] = (int16_t)(jextend(br.get(sz), sz) * (1 << sc.Al));
            k++;
        }
        return nullptr;
    }
    // AC refinement: one more bit for every coefficient that is already non-zero, and new +-1 << Al coefficients placed by runs counted over the zero ones
    const int p1 = 1 << sc.Al, m1 = -(1 << sc.Al);
    auto refine = [&](int16_t &c) {
        if (br.get(1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
    };
    int k = sc.Ss;
    if (sc.eobrun == 0) {
        for (; k <= sc.Se; k++) {
            const int rs = jdecode(br, ac), sz = rs & 15;
            int r = rs >> 4, val = 0;
            if (br.bad) return "JPEG: bad Huffman code";
            if (sz == 0) {
                if (r < 15) { sc.eobrun = 1 << r; if (r) sc.eobrun += br.get(r); break; }
            } else {
                if (sz != 1) return "JPEG: bad refinement code";
                val = br.get(1) ? p1 : m1;
            }
            for (; k <= sc.Se; k++) {
                int16_t &c = coef[JZZ[k]];
                if (c != 0) refine(c);
                else if (--r < 0) break;                      // the zero coefficient the run ends on
            }
            if (val && k <= sc.Se) coef[JZZ[k]] = (int16_t)val;
        }
    }
    if (sc.eobrun > 0) {
        for (; k <= sc.Se; k++) { int16_t &c = coef[JZZ[k]]; if (c != 0) refine(c); }
        sc.eobrun--;
    }
    return nullptr;
}

std::string load_jpeg(const uint8_t *d, size_t n, ClipImageU8 &out) {
    uint16_t qt[4][64] = {{0}};
    JHuff hdc[4], hac[4];
    std::vector<JComp> comps;
    int W = 0, H = 0, restart = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, n_scans = 0;
    bool progressive = false;
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (d[pos] != 0xff) return "JPEG: marker expected";
        const int m = d[pos + 1];
        pos += 2;
        if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01 || m == 0xff) { if (m == 0xff) pos--; continue; }
        if (m == 0xd9) break;
        if (pos + 2 > n) return "JPEG: truncated";
        const size_t len = (size_t)d[pos] << 8 | d[pos + 1];
        if (len < 2 || pos + len > n) return "JPEG: truncated segment";
        const uint8_t *s = d + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xdb) {
            size_t o = 0;
            while (o < sl) {
                const int pq = s[o] >> 4, tq = s[o] & 15;
                o++;
                if (tq > 3 || o + (size_t)64 * (pq ? 2 : 1) > sl) return "JPEG: bad quantisation table";
                for (int i = 0; i < 64; i++) { qt[tq][JZZ[i]] = pq ? (uint16_t)(s[o] << 8 | s[o + 1]) : s[o]; o += pq ? 2 : 1; }
            }
        } else if (m == 0xc4) {
            size_t o = 0;
            while (o + 17 <= sl) {
                const int tc = s[o] >> 4, th = s[o] & 15;
                if (tc > 1 || th > 3) return "JPEG: bad Huffman table";
                JHuff &h = tc ? hac[th] : hdc[th];
                int total = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[o + l]; total += h.bits[l]; }
                o += 17;
                if (total > 256 || o + (size_t)total > sl) return "JPEG: bad Huffman table";
                memcpy(h.vals, s + o, (size_t)total);
                o += (size_t)total;
                h.build();
                h.present = true;
            }
        } else if (m == 0xc0 || m == 0xc1 || m == 0xc2) {
            if (!comps.empty()) return "JPEG: more than one frame";
            if (sl < 6 || s[0] != 8) return "JPEG: only 8-bit samples are supported";
            progressive = m == 0xc2;
            H = s[1] << 8 | s[2]; W = s[3] << 8 | s[4];
            const int nc = s[5];
            if ((nc != 1 && nc != 3) || sl < (size_t)6 + 3 * nc || W <= 0 || H <= 0 || W > 16384 || H > 16384 || (int64_t)W * H > MAX_PIXELS) return "JPEG: unsupported frame";
            comps.resize((size_t)nc);
            for (int i = 0; i < nc; i++) {
                comps[i].id = s[6 + 3 * i]; comps[i].h = s[7 + 3 * i] >> 4; comps[i].v = s[7 + 3 * i] & 15; comps[i].tq = s[8 + 3 * i];
                if (comps[i].h < 1 || comps[i].h > 4 || comps[i].v < 1 || comps[i].v > 4 || comps[i].tq > 3) return "JPEG: bad component";
                hmax = std::max(hmax, comps[i].h); vmax = std::max(vmax, comps[i].v);
            }
            // (a frame of ONE component is never interleaved: its unit is a single block whatever the sampling factors say)
            if (nc == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
            mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
            // a coded block takes at least one bit of entropy data (a DC-only progressive scan of one-bit codes): a file too short for its own frame is a
            // header that asks for hundreds of MB of coefficients and planes with nothing behind it
            {
                int64_t blocks = 0;
                for (const auto &c : comps) blocks += (int64_t)mcux * c.h * mcuy * c.v;
                if (blocks > (int64_t)8 * (int64_t)n) return "JPEG: fewer bytes than the frame has blocks";
            }
            // the coefficients of the whole picture: scans (one for a baseline file as encoders write it, ten or so for a progressive one) fill them in
            for (auto &c : comps) {
                c.bw = mcux * c.h; c.bh = mcuy * c.v;
                c.stride = c.bw * 8; c.rows = c.bh * 8;
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
        } else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) {
            return "JPEG: lossless / hierarchical / arithmetic-coded files are not supported (Huffman-coded sequential and progressive DCT only)";
        } else if (m == 0xdd) {
            if (sl >= 2) restart = s[0] << 8 | s[1];
        } else if (m == 0xda) {
            if (comps.empty() || sl < 1) return "JPEG: scan before frame";
            const int ns = s[0];
            if (ns < 1 || ns > (int)comps.size() || sl < (size_t)1 + 2 * ns + 3) return "JPEG: bad scan header";
            std::vector<JComp *> sc_comps;
            for (int i = 0; i < ns; i++) {
                JComp *c = nullptr;
                for (auto &cc : comps) if (cc.id == s[1 + 2 * i]) c = &cc;
                if (!c) return "JPEG: scan names an unknown component";
                c->td = s[2 + 2 * i] >> 4; c->ta = s[2 + 2 * i] & 15;
                if (c->td > 3 || c->ta > 3) return "JPEG: bad table selector";
                c->pred = 0;
                sc_comps.push_back(c);
            }
            JScan sc;
            sc.Ss = s[1 + 2 * ns]; sc.Se = s[2 + 2 * ns]; sc.Ah = s[3 + 2 * ns] >> 4; sc.Al = s[3 + 2 * ns] & 15;
            if (progressive) {
                if (sc.Ss > sc.Se || sc.Se > 63 || sc.Ah > 13 || sc.Al > 13 || (sc.Ss == 0 && sc.Se != 0) || (sc.Ss > 0 && ns != 1)) return "JPEG: bad progressive scan parameters";
            } else if (sc.Ss != 0 || sc.Se != 63 || sc.Ah != 0 || sc.Al != 0) return "JPEG: bad scan parameters";
            for (JComp *c : sc_comps) {
                const bool need_dc = !progressive || (sc.Ss == 0 && sc.Ah == 0), need_ac = !progressive || sc.Ss > 0;
                if ((need_dc && !hdc[c->td].present) || (need_ac && !hac[c->ta].present)) return "JPEG: scan uses a missing Huffman table";
                if (!c->q_latched) { memcpy(c->q, qt[c->tq], sizeof c->q); c->q_latched = true; }     // (the table in force at the component's first scan)
            }
            JBits br{d + pos + len, d + n};
            int until_restart = restart;
            auto at_restart = [&]() {
                // the RSTn marker: already consumed by the bit reader (it feeds zeros behind a marker), or still ahead
                if (!(br.marker >= 0xd0 && br.marker <= 0xd7)) {
                    while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7)) br.p++;
                    if (br.p + 1 < br.end) br.p += 2;
                }
                br.reset();
                for (JComp *c : sc_comps) c->pred = 0;
                sc.eobrun = 0;
                until_restart = restart;
            };
            auto block = [&](JComp &c, int bx, int by) -> const char * {
                int16_t *coef = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                return progressive ? jblock_progressive(br, hdc[c.td], hac[c.ta], c.pred, coef, sc) : jblock_sequential(br, hdc[c.td], hac[c.ta], c.pred, coef);
            };
            if (ns == 1) {
                // a scan of one component is not interleaved: its blocks in raster order over the component's own extent
                JComp &c = *sc_comps[0];
                const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax, nbx = (cw + 7) / 8, nby = (chh + 7) / 8;
                for (int by = 0; by < nby; by++)
                    for (int bx = 0; bx < nbx; bx++) {
                        if (restart && until_restart == 0) at_restart();
                        if (const char *e = block(c, bx, by)) return e;
                        if (br.bad) return "JPEG: bad Huffman code";
                        if (restart) until_restart--;
                    }
            } else {
                for (int my = 0; my < mcuy; my++)
                    for (int mx = 0; mx < mcux; mx++) {
                        if (restart && until_restart == 0) at_restart();
                        for (JComp *c : sc_comps)
                            for (int by = 0; by < c->v; by++)
                                for (int bx = 0; bx < c->h; bx++)
                                    if (const char *e = block(*c, mx * c->h + bx, my * c->v + by)) return e;
                        if (br.bad) return "JPEG: bad Huffman code";
                        if (restart) until_restart--;
                    }
            }
            if (++n_scans > 64) return "JPEG: more than 64 scans";      // (encoders write about ten; every scan walks all blocks of its components)
            // the next marker segment: behind the entropy-coded bytes (stuffed zeros and restart markers belong to them)
            size_t q = pos + len;
            while (q + 1 < n && !(d[q] == 0xff && d[q + 1] != 0 && d[q + 1] != 0xff && !(d[q + 1] >= 0xd0 && d[q + 1] <= 0xd7))) q++;
            if (q + 1 >= n) break;                 // (no end-of-image marker: render what the scans gave)
            pos = q;
            continue;
        }
        pos += len;
    }
    if (n_scans == 0) return "JPEG: no image data";
    // dequantise + inverse transform, block by block
    {
        const float warm[64] = {0};
        uint8_t sink[64];
        idct8x8(warm, sink, 8);                                  // (the transform's tables are built on first use: once, before the threads)
    }
    for (auto &c : comps) {
        c.plane.assign((size_t)c.stride * c.rows, 128);
        parallel_rows(c.bh, (int64_t)c.stride * c.rows, [&](int b0, int b1) {
            float blk[64];
            for (int by = b0; by < b1; by++)
                for (int bx = 0; bx < c.bw; bx++) {
                    const int16_t *coef = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                    for (int i = 0; i < 64; i++) blk[i] = (float)((int)coef[i] * (int)c.q[i]);
                    idct8x8(blk, c.plane.data() + (size_t)(by * 8) * c.stride + (size_t)bx * 8, c.stride);
                }
        });
        std::vector<int16_t>().swap(c.coef);
    }
    // full-resolution planes.  2:1 horizontally (and vertically) subsampled components go through the triangle filter every mainstream decoder
    // applies ("fancy upsampling", libjpeg jdsample.c h2v1 / h2v2: 3/4 nearer + 1/4 farther sample per direction); other ratios are replicated
    std::vector<std::vector<uint8_t>> full(comps.size());
    for (size_t ci = 0; ci < comps.size(); ci++) {
        const JComp &c = comps[ci];
        std::vector<uint8_t> &f = full[ci];
        f.resize((size_t)W * H);
        const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax;      // valid extent of the component
        const bool h2 = hmax == 2 * c.h, v1 = vmax == c.v, v2 = vmax == 2 * c.v;
        if (h2 && (v1 || v2) && cw >= 1) {
            parallel_rows(H, (int64_t)W * H, [&](int y0, int y1) {
            for (int y = y0; y < y1; y++) {
                const int r = v2 ? y >> 1 : y;
                int rf = v2 ? ((y & 1) ? r + 1 : r - 1) : r;                                 // the farther row of the pair
                if (rf < 0) rf = 0;
                if (rf > chh - 1) rf = chh - 1;
                const uint8_t *nr = c.plane.data() + (size_t)std::min(r, chh - 1) * c.stride, *fr = c.plane.data() + (size_t)rf * c.stride;
                uint8_t *o = f.data() + (size_t)y * W;
                auto colsum = [&](int i) { return v2 ? 3 * (int)nr[i] + (int)fr[i] : 4 * (int)nr[i]; };      // (x 4 of the vertical blend)
                for (int i = 0; i < cw; i++) {
                    const int cur = colsum(i), prev = colsum(i > 0 ? i - 1 : 0), next = colsum(i + 1 < cw ? i + 1 : cw - 1);
                    const int x0 = 2 * i, x1 = 2 * i + 1;
                    const int a = i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + prev + 8) >> 4;
                    const int b = i == cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + next + 7) >> 4;
                    if (x0 < W) o[x0] = (uint8_t)a;
                    if (x1 < W) o[x1] = (uint8_t)b;
                }
            }
            });
        } else {
            parallel_rows(H, (int64_t)W * H, [&](int y0, int y1) {
                for (int y = y0; y < y1; y++)
                    for (int x = 0; x < W; x++) f[(size_t)y * W + x] = c.plane[(size_t)(y * c.v / vmax) * c.stride + (size_t)(x * c.h / hmax)];
            });
        }
    }
    out.nx = W; out.ny = H; out.rgb.resize((size_t)3 * W * H);
    parallel_rows(H, (int64_t)W * H, [&](int y0, int y1) {
        for (size_t i = (size_t)y0 * W; i < (size_t)y1 * W; i++) {
            uint8_t *o = out.rgb.data() + 3 * i;
            if (comps.size() == 1) { o[0] = o[1] = o[2] = full[0][i]; continue; }
            const float Y = (float)full[0][i], cb = (float)full[1][i] - 128.0f, cr = (float)full[2][i] - 128.0f;
            const float rgb[3] = {Y + 1.402f * cr, Y - 0.344136f * cb - 0.714136f * cr, Y + 1.772f * cb};
            for (int k = 0; k < 3; k++) o[k] = to_byte(rgb[k]);
        }
    });
    return "";
}

}  // namespace

std::string clip_image_load_from_bytes(const uint8_t *d, size_t n, ClipImageU8 &out) {
    out = ClipImageU8();
    if (!d || n < 8) return "image: too few bytes";
    static const uint8_t png_sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (!memcmp(d, png_sig, 8)) return load_png(d, n, out);
    if (d[0] == 0xff && d[1] == 0xd8) return load_jpeg(d, n, out);
    if (d[0] == 'B' && d[1] == 'M') return load_bmp(d, n, out);
    if (d[0] == 'P' && (d[1] == '5' || d[1] == '6')) return load_pnm(d, n, out);
    return "image: unknown format (PNG, JPEG, BMP and binary PNM are read)";
}

}  // namespace mi355
