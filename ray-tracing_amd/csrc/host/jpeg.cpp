// Baseline (sequential, Huffman, 8-bit) JPEG decoder and binary PPM reader producing 8-bit RGB.
// The reference decodes image textures with the `image` crate (src/json.rs:349-354); that
// decoder is not in the reference tree, so exact texel parity with it is unpinned.  This decoder
// uses the classic 13-bit fixed-point Loeffler/Ligtenberg/Moschytz inverse DCT and the JFIF
// YCbCr conversion; chroma planes are upsampled by replication (the `image` crate interpolates subsampled chroma, so
// texels of a chroma-subsampled file may differ from the reference's by a few levels along colour edges).  Every index
// and bit count taken from the file is checked before use: a malformed file is an exception, never an out-of-bounds access.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "scene.hpp"

namespace v4ray {

namespace {

struct Huff {
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    // canonical decode tables
    int mincode[17], maxcode[18], valptr[17];
    bool present = false;
    void build() {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; ++l) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        present = true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int dc_pred = 0;
    int blocks_w = 0, blocks_h = 0;  // allocated blocks (MCU aligned)
    std::vector<uint8_t> plane;      // blocks_w*8 x blocks_h*8
};

struct BitReader {
    const uint8_t *p, *end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    BitReader(const uint8_t *b, const uint8_t *e) : p(b), end(e) {}
    void fill() {
        while (nbits <= 24) {
            uint32_t byte = 0;
            if (!hit_marker && p < end) {
                byte = *p;
                if (byte == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else {
                        hit_marker = true;  // leave the marker in place; feed zeros
                        byte = 0;
                    }
                } else {
                    ++p;
                }
            } else {
                hit_marker = true;
            }
            acc |= byte << (24 - nbits);
            nbits += 8;
        }
    }
    int bit() {
        if (nbits < 1) fill();
        int b = (int)(acc >> 31);
        acc <<= 1;
        --nbits;
        return b;
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
    void reset() {
        acc = 0;
        nbits = 0;
        hit_marker = false;
    }
};

int decode_symbol(BitReader &br, const Huff &h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
        code = (code << 1) | br.bit();
        if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    throw std::runtime_error("JPEG: bad Huffman code");
}

int extend(int v, int n) { return (n && v < (1 << (n - 1))) ? v - (1 << n) + 1 : v; }

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// 64-bit intermediates: coefficients of a crafted file cannot overflow them
typedef long long wide;
inline wide fx(double x) { return (wide)(x * 4096 + 0.5); }
inline uint8_t clamp8(wide x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

struct Idct1D {
    wide x0, x1, x2, x3, t0, t1, t2, t3;
    Idct1D(wide s0, wide s1, wide s2, wide s3, wide s4, wide s5, wide s6, wide s7) {
        wide p1, p2, p3, p4, p5;
        p2 = s2;
        p3 = s6;
        p1 = (p2 + p3) * fx(0.5411961);
        wide e2 = p1 + p3 * fx(-1.847759065);
        wide e3 = p1 + p2 * fx(0.765366865);
        p2 = s0;
        p3 = s4;
        wide e0 = (p2 + p3) * 4096;
        wide e1 = (p2 - p3) * 4096;
        x0 = e0 + e3;
        x3 = e0 - e3;
        x1 = e1 + e2;
        x2 = e1 - e2;
        t0 = s7;
        t1 = s5;
        t2 = s3;
        t3 = s1;
        p3 = t0 + t2;
        p4 = t1 + t3;
        p1 = t0 + t3;
        p2 = t1 + t2;
        p5 = (p3 + p4) * fx(1.175875602);
        t0 = t0 * fx(0.298631336);
        t1 = t1 * fx(2.053119869);
        t2 = t2 * fx(3.072711026);
        t3 = t3 * fx(1.501321110);
        p1 = p5 + p1 * fx(-0.899976223);
        p2 = p5 + p2 * fx(-2.562915447);
        p3 = p3 * fx(-1.961570560);
        p4 = p4 * fx(-0.390180644);
        t3 += p1 + p4;
        t2 += p2 + p3;
        t1 += p2 + p4;
        t0 += p1 + p3;
    }
};

void idct_block(const int *coef, uint8_t *out, int stride) {
    wide tmp[64];
    for (int c = 0; c < 8; ++c) {
        const int *d = coef + c;
        if (!d[8] && !d[16] && !d[24] && !d[32] && !d[40] && !d[48] && !d[56]) {
            wide dc = (wide)d[0] * 4;
            for (int r = 0; r < 8; ++r) tmp[r * 8 + c] = dc;
            continue;
        }
        Idct1D k(d[0], d[8], d[16], d[24], d[32], d[40], d[48], d[56]);
        wide x0 = k.x0 + 512, x1 = k.x1 + 512, x2 = k.x2 + 512, x3 = k.x3 + 512;
        tmp[0 * 8 + c] = (x0 + k.t3) >> 10;
        tmp[7 * 8 + c] = (x0 - k.t3) >> 10;
        tmp[1 * 8 + c] = (x1 + k.t2) >> 10;
        tmp[6 * 8 + c] = (x1 - k.t2) >> 10;
        tmp[2 * 8 + c] = (x2 + k.t1) >> 10;
        tmp[5 * 8 + c] = (x2 - k.t1) >> 10;
        tmp[3 * 8 + c] = (x3 + k.t0) >> 10;
        tmp[4 * 8 + c] = (x3 - k.t0) >> 10;
    }
    for (int r = 0; r < 8; ++r) {
        const wide *v = tmp + r * 8;
        uint8_t *o = out + r * stride;
        Idct1D k(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const wide bias = 65536 + (128 << 17);
        wide x0 = k.x0 + bias, x1 = k.x1 + bias, x2 = k.x2 + bias, x3 = k.x3 + bias;
        o[0] = clamp8((x0 + k.t3) >> 17);
        o[7] = clamp8((x0 - k.t3) >> 17);
        o[1] = clamp8((x1 + k.t2) >> 17);
        o[6] = clamp8((x1 - k.t2) >> 17);
        o[2] = clamp8((x2 + k.t1) >> 17);
        o[5] = clamp8((x2 - k.t1) >> 17);
        o[3] = clamp8((x3 + k.t0) >> 17);
        o[4] = clamp8((x3 - k.t0) >> 17);
    }
}

uint16_t be16(const uint8_t *p) { return (uint16_t)((p[0] << 8) | p[1]); }

}  // namespace

Image decode_jpeg(const uint8_t *data, size_t size) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw std::runtime_error("JPEG: missing SOI");
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int width = 0, height = 0, hmax = 1, vmax = 1, restart_interval = 0;
    bool have_frame = false;
    size_t pos = 2;
    while (true) {
        if (pos + 4 > size) throw std::runtime_error("JPEG: truncated before scan");
        if (data[pos] != 0xFF) throw std::runtime_error("JPEG: expected marker");
        while (pos < size && data[pos] == 0xFF) ++pos;
        if (pos >= size) throw std::runtime_error("JPEG: truncated at a marker");
        uint8_t m = data[pos++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) throw std::runtime_error("JPEG: EOI before scan");
        if (pos + 2 > size) throw std::runtime_error("JPEG: truncated segment");
        size_t len = be16(data + pos);
        if (len < 2 || pos + len > size) throw std::runtime_error("JPEG: bad segment length");
        const uint8_t *seg = data + pos + 2;
        size_t n = len - 2;
        if (m == 0xDB) {  // DQT
            size_t i = 0;
            while (i < n) {
                int pq = seg[i] >> 4, tq = seg[i] & 15;
                ++i;
                if (tq > 3) throw std::runtime_error("JPEG: bad quant table id");
                for (int k = 0; k < 64; ++k) {
                    if (i + (pq ? 2 : 1) > n) throw std::runtime_error("JPEG: truncated DQT");
                    qt[tq][ZIGZAG[k]] = pq ? be16(seg + i) : seg[i];
                    i += pq ? 2 : 1;
                }
                qt_present[tq] = true;
            }
        } else if (m == 0xC4) {  // DHT
            size_t i = 0;
            while (i < n) {
                if (i + 17 > n) throw std::runtime_error("JPEG: truncated DHT");
                int tc = seg[i] >> 4, th = seg[i] & 15;
                if (th > 3 || tc > 1) throw std::runtime_error("JPEG: bad Huffman table id");
                Huff &h = tc ? ac[th] : dc[th];
                int total = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) {
                    h.bits[l] = seg[i + l];
                    total += h.bits[l];
                }
                i += 17;
                if (total > 256 || i + total > n) throw std::runtime_error("JPEG: bad DHT");
                std::memcpy(h.vals, seg + i, (size_t)total);
                i += total;
                h.build();
            }
        } else if (m == 0xC0 || m == 0xC1) {  // SOF0/1 baseline / extended sequential
            if (n < 6 || seg[0] != 8) throw std::runtime_error("JPEG: only 8-bit precision is supported");
            height = be16(seg + 1);
            width = be16(seg + 3);
            int nc = seg[5];
            if ((nc != 1 && nc != 3) || n < (size_t)(6 + 3 * nc)) throw std::runtime_error("JPEG: unsupported component count");
            comps.resize(nc);
            for (int c = 0; c < nc; ++c) {
                comps[c].id = seg[6 + 3 * c];
                comps[c].h = seg[7 + 3 * c] >> 4;
                comps[c].v = seg[7 + 3 * c] & 15;
                comps[c].tq = seg[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3)
                    throw std::runtime_error("JPEG: bad component spec");
                hmax = std::max(hmax, comps[c].h);
                vmax = std::max(vmax, comps[c].v);
            }
            if (width <= 0 || height <= 0) throw std::runtime_error("JPEG: bad dimensions");
            if ((long long)width * height > (1ll << 28)) throw std::runtime_error("JPEG: more than 2^28 pixels");
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            throw std::runtime_error("JPEG: only baseline/sequential Huffman JPEG is supported");
        } else if (m == 0xDD) {
            if (n < 2) throw std::runtime_error("JPEG: bad DRI");
            restart_interval = be16(seg);
        } else if (m == 0xDA) {  // SOS
            if (!have_frame) throw std::runtime_error("JPEG: scan before frame header");
            if (n < 1) throw std::runtime_error("JPEG: truncated SOS");
            int ns = seg[0];
            if (ns != (int)comps.size()) throw std::runtime_error("JPEG: non-interleaved scans are not supported");
            if ((size_t)(1 + 2 * ns) > n) throw std::runtime_error("JPEG: truncated SOS");
            for (int s = 0; s < ns; ++s) {
                int cid = seg[1 + 2 * s];
                bool found = false;
                for (auto &c : comps)
                    if (c.id == cid) {
                        c.td = seg[2 + 2 * s] >> 4;
                        c.ta = seg[2 + 2 * s] & 15;
                        if (c.td > 3 || c.ta > 3) throw std::runtime_error("JPEG: bad Huffman table id in scan");
                        found = true;
                    }
                if (!found) throw std::runtime_error("JPEG: scan names an unknown component");
            }
            pos += len;
            break;
        }
        pos += len;
    }
    int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
    int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
    for (auto &c : comps) {
        if (!qt_present[c.tq] || !dc[c.td].present || !ac[c.ta].present) throw std::runtime_error("JPEG: missing table");
        c.blocks_w = mcus_x * c.h;
        c.blocks_h = mcus_y * c.v;
        c.plane.assign((size_t)c.blocks_w * 8 * c.blocks_h * 8, 0);
    }
    BitReader br(data + pos, data + size);
    int until_restart = restart_interval;
    int coef[64];
    for (int my = 0; my < mcus_y; ++my)
        for (int mx = 0; mx < mcus_x; ++mx) {
            if (restart_interval && until_restart == 0) {
                // skip to the RSTn marker and reset predictors
                const uint8_t *p = br.p;
                while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) ++p;
                if (p + 1 < br.end) p += 2;
                br.p = p;
                br.reset();
                for (auto &c : comps) c.dc_pred = 0;
                until_restart = restart_interval;
            }
            for (auto &c : comps)
                for (int by = 0; by < c.v; ++by)
                    for (int bx = 0; bx < c.h; ++bx) {
                        std::memset(coef, 0, sizeof coef);
                        int t = decode_symbol(br, dc[c.td]);
                        if (t > 11) throw std::runtime_error("JPEG: bad DC size");  // 8-bit samples: differences need <= 11 bits
                        int diff = t ? extend(br.bits(t), t) : 0;
                        c.dc_pred = std::max(-32768, std::min(32767, c.dc_pred + diff));
                        coef[0] = c.dc_pred * qt[c.tq][0];
                        for (int k = 1; k < 64;) {
                            int rs = decode_symbol(br, ac[c.ta]);
                            int r = rs >> 4, s = rs & 15;
                            if (s == 0) {
                                if (r != 15) break;
                                k += 16;
                                continue;
                            }
                            if (s > 10) throw std::runtime_error("JPEG: bad AC size");  // 8-bit samples: <= 10 bits
                            k += r;
                            if (k > 63) throw std::runtime_error("JPEG: coefficient overrun");
                            int z = ZIGZAG[k];
                            coef[z] = extend(br.bits(s), s) * qt[c.tq][z];
                            ++k;
                        }
                        int stride = c.blocks_w * 8;
                        uint8_t *dst = c.plane.data() + (size_t)(my * c.v + by) * 8 * stride + (size_t)(mx * c.h + bx) * 8;
                        idct_block(coef, dst, stride);
                    }
            if (restart_interval) --until_restart;
        }
    Image img;
    img.width = (uint32_t)width;
    img.height = (uint32_t)height;
    img.rgb.resize((size_t)width * height * 3);
    auto sample = [&](const Component &c, int x, int y) {
        int sx = x * c.h / hmax, sy = y * c.v / vmax;
        return c.plane[(size_t)sy * c.blocks_w * 8 + sx];
    };
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            uint8_t *o = &img.rgb[((size_t)y * width + x) * 3];
            if (comps.size() == 1) {
                o[0] = o[1] = o[2] = sample(comps[0], x, y);
            } else {
                float Y = (float)sample(comps[0], x, y);
                float cb = (float)sample(comps[1], x, y) - 128.0f;
                float cr = (float)sample(comps[2], x, y) - 128.0f;
                o[0] = clamp8((int)(Y + 1.40200f * cr + 0.5f));
                o[1] = clamp8((int)(Y - 0.34414f * cb - 0.71414f * cr + 0.5f));
                o[2] = clamp8((int)(Y + 1.77200f * cb + 0.5f));
            }
        }
    return img;
}

static Image decode_ppm(const std::string &bytes) {
    std::istringstream in(bytes);
    std::string magic;
    in >> magic;
    if (magic != "P6") throw std::runtime_error("PPM: only binary P6 is supported");
    auto next_int = [&]() {
        while (true) {
            int c = in.peek();
            if (c == '#') {
                std::string skip;
                std::getline(in, skip);
            } else if (c == ' ' || c == '\n' || c == '\r' || c == '\t') {
                in.get();
            } else {
                break;
            }
        }
        int v = -1;
        in >> v;
        return v;
    };
    int w = next_int(), h = next_int(), maxv = next_int();
    if (w <= 0 || h <= 0 || maxv != 255) throw std::runtime_error("PPM: bad header");
    in.get();
    Image img;
    img.width = (uint32_t)w;
    img.height = (uint32_t)h;
    img.rgb.resize((size_t)w * h * 3);
    in.read((char *)img.rgb.data(), (std::streamsize)img.rgb.size());
    if ((size_t)in.gcount() != img.rgb.size()) throw std::runtime_error("PPM: truncated");
    return img;
}

Image load_image(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("failed to open image: " + path);  // json.rs:351
    std::stringstream ss;
    ss << f.rdbuf();
    std::string bytes = ss.str();
    if (bytes.size() >= 2 && (uint8_t)bytes[0] == 0xFF && (uint8_t)bytes[1] == 0xD8)
        return decode_jpeg((const uint8_t *)bytes.data(), bytes.size());
    if (bytes.size() >= 2 && bytes[0] == 'P' && bytes[1] == '6') return decode_ppm(bytes);
    throw std::runtime_error("failed to decode image (JPEG or binary PPM expected): " + path);  // json.rs:353
}

}  // namespace v4ray
