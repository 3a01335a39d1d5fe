// JPEG decoder for texture ingest (ITU-T T.81): Huffman coded, 8-bit, baseline/extended sequential (SOF0/SOF1)
// and progressive (SOF2) frames, interleaved and non-interleaved scans, restart intervals, one or three
// components with sampling factors 1 or 2 (others are replicated), JFIF YCbCr or Adobe RGB.
//
// The reference decodes textures through sf::Image (image_loader.h:8,52), whose JPEG reader is not part of the
// reference tree; decoders differ in the last bit of the inverse transform and in chroma interpolation, so this
// one is checked against libjpeg's output within a few grey levels, not bit for bit. It uses an exact
// (double precision, separable) inverse DCT and libjpeg's triangle-filter chroma upsampling.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "pt_internal.h"

namespace ptamd {

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    uint16_t look[512];         // 9-bit prefix -> (length << 8 | value), 0 = longer code
    // false when the code lengths over-subscribe the code space (more codes of some length than prefixes
    // are left: the Kraft inequality), which would index past the tables below
    bool build(const uint8_t* bits, const uint8_t* v, int n)
    {
        for (int len = 1, code = 0; len <= 16; len++) {
            if (code + (int)bits[len - 1] > (1 << len)) return false;
            code = (code + (int)bits[len - 1]) << 1;
        }
        present = true;
        memcpy(vals, v, (size_t)n);
        int code = 0, k = 0;
        memset(look, 0, sizeof(look));
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k; mincode[len] = code;
            for (int i = 0; i < bits[len - 1]; i++, k++, code++) {
                if (len <= 9) {
                    int first = code << (9 - len);
                    for (int j = 0; j < (1 << (9 - len)); j++) look[first + j] = (uint16_t)(len << 8 | vals[k]);
                }
            }
            maxcode[len] = bits[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int blocks_w = 0, blocks_h = 0;     // allocated (whole MCUs)
    int width = 0, height = 0;          // samples that carry image data
    std::vector<int16_t> coef;          // blocks_w * blocks_h * 64, natural order
    int dc_pred = 0, td = 0, ta = 0;
    std::vector<uint8_t> plane;         // blocks_w*8 x blocks_h*8
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int n = 0;
    bool hit_marker = false;
    void fill()
    {
        while (n <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0x00) p += 2;
                    else { hit_marker = true; b = 0; }       // a marker: feed zeros until the caller deals with it
                } else p++;
            }
            acc |= (uint32_t)b << (24 - n);
            n += 8;
        }
    }
    int peek(int k) { if (n < k) fill(); return (int)(acc >> (32 - k)); }
    void skip(int k) { acc <<= k; n -= k; }
    int bits(int k) { if (k == 0) return 0; int v = peek(k); skip(k); return v; }
    int bit() { return bits(1); }
    void reset() { acc = 0; n = 0; hit_marker = false; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Decoder {
    const uint8_t* data; size_t size;
    int W = 0, H = 0, hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    bool progressive = false, have_frame = false;
    int adobe_transform = -1;
    uint16_t qt[4][64]; bool have_qt[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    std::vector<Component> comps;
    int restart_interval = 0;
    int eobrun = 0;
    const char* err = nullptr;

    int decode_symbol(BitReader& br, const Huff& h)
    {
        int look = h.look[br.peek(9)];
        if (look) { br.skip(look >> 8); return look & 0xff; }
        int code = br.peek(16);
        for (int len = 10; len <= 16; len++) {
            int c = code >> (16 - len);
            if (h.maxcode[len] >= 0 && c <= h.maxcode[len] && c >= h.mincode[len]) { br.skip(len); return h.vals[h.valptr[len] + c - h.mincode[len]]; }
        }
        err = "bad Huffman code";
        br.skip(16);
        return 0;
    }

    void block_sequential(BitReader& br, Component& c, int16_t* b)
    {
        int t = decode_symbol(br, dc[c.td]);
        if (t > 15) { err = "bad DC category"; return; }
        int diff = t ? extend(br.bits(t), t) : 0;
        c.dc_pred += diff;
        b[0] = (int16_t)c.dc_pred;
        const Huff& h = ac[c.ta];
        for (int k = 1; k < 64;) {
            int rs = decode_symbol(br, h), r = rs >> 4, s = rs & 15;
            if (s == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) { err = "coefficient index out of range"; return; }
            b[kZigzag[k]] = (int16_t)extend(br.bits(s), s);
            k++;
        }
    }
    void block_dc_first(BitReader& br, Component& c, int16_t* b, int al)
    {
        int t = decode_symbol(br, dc[c.td]);
        if (t > 15) { err = "bad DC category"; return; }
        int diff = t ? extend(br.bits(t), t) : 0;
        c.dc_pred += diff;
        b[0] = (int16_t)(c.dc_pred * (1 << al));
    }
    void block_dc_refine(BitReader& br, int16_t* b, int al) { if (br.bit()) b[0] |= (int16_t)(1 << al); }
    void block_ac_first(BitReader& br, Component& c, int16_t* b, int ss, int se, int al)
    {
        if (eobrun > 0) { eobrun--; return; }
        const Huff& h = ac[c.ta];
        for (int k = ss; k <= se;) {
            int rs = decode_symbol(br, h), r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) { eobrun = (1 << r) - 1; if (r) eobrun += br.bits(r); break; }
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) { err = "coefficient index out of range"; return; }
            b[kZigzag[k]] = (int16_t)(extend(br.bits(s), s) * (1 << al));
            k++;
        }
    }
    void block_ac_refine(BitReader& br, Component& c, int16_t* b, int ss, int se, int al)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        const Huff& h = ac[c.ta];
        int k = ss;
        auto refine = [&](int16_t& v) {
            if (br.bit() && (v & p1) == 0) v = (int16_t)(v + (v >= 0 ? p1 : m1));
        };
        if (eobrun == 0) {
            for (; k <= se; k++) {
                int rs = decode_symbol(br, h), r = rs >> 4, s = rs & 15, val = 0;
                if (s == 0) {
                    if (r < 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
                    // r == 15: sixteen zero coefficients to pass
                } else val = br.bit() ? p1 : m1;
                for (; k <= se; k++) {
                    int16_t& v = b[kZigzag[k]];
                    if (v != 0) refine(v);
                    else if (--r < 0) break;
                }
                if (s && k <= se) b[kZigzag[k]] = (int16_t)val;
                if (err) return;
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) { int16_t& v = b[kZigzag[k]]; if (v != 0) refine(v); }
            eobrun--;
        }
    }

    bool parse_frame(const uint8_t* p, int len, bool prog)
    {
        if (len < 6) return false;
        if (p[0] != 8) { err = "only 8-bit samples are supported"; return false; }
        H = p[1] << 8 | p[2]; W = p[3] << 8 | p[4];
        int n = p[5];
        if (W <= 0 || H <= 0 || (uint64_t)W * H > (1ull << 28)) { err = "bad image size"; return false; }
        if (!(n == 1 || n == 3) || len < 6 + 3 * n) { err = "only one- and three-component images are supported"; return false; }
        comps.assign((size_t)n, Component());
        for (int i = 0; i < n; i++) {
            Component& c = comps[(size_t)i];
            c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i] & 3;
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) { err = "bad sampling factor"; return false; }
            hmax = c.h > hmax ? c.h : hmax; vmax = c.v > vmax ? c.v : vmax;
        }
        if (n == 1) { comps[0].h = comps[0].v = 1; hmax = vmax = 1; }
        mcus_x = (W + 8 * hmax - 1) / (8 * hmax); mcus_y = (H + 8 * vmax - 1) / (8 * vmax);
        for (auto& c : comps) {
            c.blocks_w = mcus_x * c.h; c.blocks_h = mcus_y * c.v;
            c.width = (W * c.h + hmax - 1) / hmax; c.height = (H * c.v + vmax - 1) / vmax;
            c.coef.assign((size_t)c.blocks_w * c.blocks_h * 64, 0);
        }
        progressive = prog; have_frame = true;
        return true;
    }

    // entropy-coded segment after an SOS header; returns the position of the marker that ends it
    size_t decode_scan(size_t pos, const std::vector<int>& sc, int ss, int se, int ah, int al)
    {
        BitReader br; br.p = data + pos; br.end = data + size;
        for (int ci : sc) comps[(size_t)ci].dc_pred = 0;
        eobrun = 0;
        const bool interleaved = sc.size() > 1;
        int total;
        Component& c0 = comps[(size_t)sc[0]];
        int bw0 = (c0.width + 7) / 8, bh0 = (c0.height + 7) / 8;
        if (interleaved) total = mcus_x * mcus_y; else total = bw0 * bh0;
        auto one_block = [&](Component& c, int bx, int by) {
            int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            if (!progressive) block_sequential(br, c, b);
            else if (ss == 0) { if (ah == 0) block_dc_first(br, c, b, al); else block_dc_refine(br, b, al); }
            else { if (ah == 0) block_ac_first(br, c, b, ss, se, al); else block_ac_refine(br, c, b, ss, se, al); }
        };
        int until_restart = restart_interval;
        for (int m = 0; m < total && !err; m++) {
            if (restart_interval && until_restart == 0) {
                // byte-align, expect RSTn
                br.reset();
                const uint8_t* q = br.p;
                while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
                    if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) break;     // some other marker: the data ended early
                    q++;
                }
                if (q + 1 < br.end && q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7) br.p = q + 2;
                else { err = "restart marker missing"; break; }
                for (int ci : sc) comps[(size_t)ci].dc_pred = 0;
                eobrun = 0;
                until_restart = restart_interval;
            }
            if (interleaved) {
                int mx = m % mcus_x, my = m / mcus_x;
                for (int ci : sc) {
                    Component& c = comps[(size_t)ci];
                    for (int v = 0; v < c.v; v++) for (int h = 0; h < c.h; h++) one_block(c, mx * c.h + h, my * c.v + v);
                }
            } else one_block(c0, m % bw0, m / bw0);
            if (restart_interval) until_restart--;
        }
        // the next marker
        const uint8_t* q = br.p;
        if (br.hit_marker) return (size_t)(q - data);
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
        return (size_t)(q - data);
    }

    void reconstruct(Component& c)
    {
        static double cs[8][8];
        static bool init = false;
        if (!init) {
            for (int x = 0; x < 8; x++) for (int u = 0; u < 8; u++) cs[x][u] = (u == 0 ? std::sqrt(0.125) : 0.5) * std::cos((2 * x + 1) * u * M_PI / 16.0);
            init = true;
        }
        const uint16_t* q = qt[c.tq];
        const int pw = c.blocks_w * 8;
        c.plane.assign((size_t)pw * c.blocks_h * 8, 0);
        for (int by = 0; by < c.blocks_h; by++) for (int bx = 0; bx < c.blocks_w; bx++) {
            const int16_t* b = &c.coef[((size_t)by * c.blocks_w + bx) * 64];
            double f[64], t[64];
            bool ac_zero = true;
            for (int i = 0; i < 64; i++) { f[i] = (double)b[i] * (double)q[i]; if (i && b[i]) ac_zero = false; }
            uint8_t* out = &c.plane[(size_t)by * 8 * pw + (size_t)bx * 8];
            if (ac_zero) {
                double v = f[0] * 0.125 + 128.0;
                int iv = (int)std::floor(v + 0.5); iv = iv < 0 ? 0 : iv > 255 ? 255 : iv;
                for (int y = 0; y < 8; y++) memset(out + (size_t)y * pw, iv, 8);
                continue;
            }
            for (int v = 0; v < 8; v++) for (int x = 0; x < 8; x++) {        // rows: t[v][x] = sum_u cs[x][u] f[v][u]
                double s = 0;
                for (int u = 0; u < 8; u++) s += cs[x][u] * f[v * 8 + u];
                t[v * 8 + x] = s;
            }
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                double s = 0;
                for (int v = 0; v < 8; v++) s += cs[y][v] * t[v * 8 + x];
                int iv = (int)std::floor(s + 128.0 + 0.5);
                out[(size_t)y * pw + x] = (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
            }
        }
    }

    // component plane -> W x H samples (libjpeg's "fancy" triangle filters for factors of two, replication otherwise)
    void upsample(const Component& c, std::vector<uint8_t>& out)
    {
        const int pw = c.blocks_w * 8, cw = c.width, ch = c.height;
        const int fx = hmax / c.h, fy = vmax / c.v;
        out.assign((size_t)W * H, 0);
        auto in = [&](int x, int y) -> int { x = x < 0 ? 0 : x >= cw ? cw - 1 : x; y = y < 0 ? 0 : y >= ch ? ch - 1 : y; return c.plane[(size_t)y * pw + x]; };
        if (fx == 1 && fy == 1 && hmax % c.h == 0 && vmax % c.v == 0) {
            for (int y = 0; y < H; y++) memcpy(&out[(size_t)y * W], &c.plane[(size_t)y * pw], (size_t)W);
        } else if (fx == 2 && fy == 1 && hmax == 2 * c.h && vmax == c.v) {
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
                int i = x >> 1, v;
                if (cw == 1) v = in(0, y);
                else if (x == 0) v = in(0, y);
                else if (x == 2 * cw - 1) v = in(cw - 1, y);
                else if (x & 1) v = (3 * in(i, y) + in(i + 1, y) + 2) >> 2;
                else v = (3 * in(i, y) + in(i - 1, y) + 1) >> 2;
                out[(size_t)y * W + x] = (uint8_t)v;
            }
        } else if (fx == 2 && fy == 2 && hmax == 2 * c.h && vmax == 2 * c.v) {
            for (int y = 0; y < H; y++) {
                int j = y >> 1, jn = (y & 1) ? j + 1 : j - 1;       // nearer row j, farther row jn (edge rows replicate)
                auto colsum = [&](int i) { return 3 * in(i, j) + in(i, jn); };
                for (int x = 0; x < W; x++) {
                    int i = x >> 1, v;
                    int cur = colsum(i);
                    if (cw == 1) v = (cur * 4 + 8) >> 4;
                    else if (x == 0) v = (cur * 4 + 8) >> 4;
                    else if (x == 2 * cw - 1) v = (cur * 4 + 7) >> 4;
                    else if (x & 1) v = (cur * 3 + colsum(i + 1) + 7) >> 4;
                    else v = (cur * 3 + colsum(i - 1) + 8) >> 4;
                    out[(size_t)y * W + x] = (uint8_t)v;
                }
            }
        } else {
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) out[(size_t)y * W + x] = (uint8_t)in(x * c.h / hmax, y * c.v / vmax);
        }
    }

    bool run(int& w_out, int& h_out, std::vector<uint8_t>& rgb)
    {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { err = "not a JPEG file"; return false; }
        size_t pos = 2;
        bool done = false;
        while (!done && pos + 4 <= size) {
            if (data[pos] != 0xFF) { pos++; continue; }
            int m = data[pos + 1];
            if (m == 0xFF) { pos++; continue; }
            if (m == 0xD9) { done = true; break; }
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0x00) { pos += 2; continue; }
            int len = data[pos + 2] << 8 | data[pos + 3];
            if (len < 2 || pos + 2 + (size_t)len > size) { err = "truncated segment"; return false; }
            const uint8_t* p = data + pos + 4;
            int n = len - 2;
            if (m == 0xDB) {
                for (int i = 0; i < n;) {
                    int pq = p[i] >> 4, tq = p[i] & 15; i++;
                    if (tq > 3 || i + (pq ? 128 : 64) > n) { err = "bad quantization table"; return false; }
                    for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? (uint16_t)(p[i] << 8 | p[i + 1]) : p[i]; i += pq ? 2 : 1; }
                    have_qt[tq] = true;
                }
            } else if (m == 0xC4) {
                for (int i = 0; i < n;) {
                    if (i + 17 > n) { err = "bad Huffman table"; return false; }
                    int tc = p[i] >> 4, th = p[i] & 15, cnt = 0;
                    for (int k = 0; k < 16; k++) cnt += p[i + 1 + k];
                    if (tc > 1 || th > 3 || cnt > 256 || i + 17 + cnt > n) { err = "bad Huffman table"; return false; }
                    if (!(tc ? ac[th] : dc[th]).build(p + i + 1, p + i + 17, cnt)) { err = "bad Huffman table"; return false; }
                    i += 17 + cnt;
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) { err = "more than one frame"; return false; }
                if (!parse_frame(p, n, m == 0xC2)) { if (!err) err = "bad frame header"; return false; }
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
                err = "lossless, hierarchical and arithmetic-coded JPEG are not supported"; return false;
            } else if (m == 0xDD) {
                if (n >= 2) restart_interval = p[0] << 8 | p[1];
            } else if (m == 0xEE) {
                if (n >= 12 && !memcmp(p, "Adobe", 5)) adobe_transform = p[11];
            } else if (m == 0xDA) {
                if (!have_frame) { err = "scan before frame header"; return false; }
                int ns = p[0];
                if (ns < 1 || ns > (int)comps.size() || n < 1 + 2 * ns + 3) { err = "bad scan header"; return false; }
                std::vector<int> sc;
                for (int i = 0; i < ns; i++) {
                    int id = p[1 + 2 * i], ci = -1;
                    for (size_t k = 0; k < comps.size(); k++) if (comps[k].id == id) ci = (int)k;
                    if (ci < 0) { err = "scan names an unknown component"; return false; }
                    comps[(size_t)ci].td = p[2 + 2 * i] >> 4; comps[(size_t)ci].ta = p[2 + 2 * i] & 15;
                    if (comps[(size_t)ci].td > 3 || comps[(size_t)ci].ta > 3) { err = "bad table selector"; return false; }
                    sc.push_back(ci);
                }
                int ss = p[1 + 2 * ns], se = p[2 + 2 * ns], ah = p[3 + 2 * ns] >> 4, al = p[3 + 2 * ns] & 15;
                if (!progressive) { ss = 0; se = 63; ah = al = 0; }
                else if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1) || al > 13) { err = "bad progressive scan parameters"; return false; }
                for (int ci : sc) {
                    const Component& c = comps[(size_t)ci];
                    bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss > 0;
                    if ((need_dc && !dc[c.td].present) || (need_ac && !ac[c.ta].present)) { err = "scan uses an undefined Huffman table"; return false; }
                }
                pos = decode_scan(pos + 2 + (size_t)len, sc, ss, se, ah, al);
                if (err) return false;
                continue;
            }
            pos += 2 + (size_t)len;
        }
        if (!have_frame) { err = "no frame header"; return false; }
        for (auto& c : comps) { if (!have_qt[c.tq]) { err = "undefined quantization table"; return false; } reconstruct(c); }
        w_out = W; h_out = H;
        rgb.assign((size_t)W * H * 3, 0);
        if (comps.size() == 1) {
            std::vector<uint8_t> y; upsample(comps[0], y);
            for (size_t i = 0; i < (size_t)W * H; i++) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = y[i];
            return true;
        }
        std::vector<uint8_t> a, b, c;
        upsample(comps[0], a); upsample(comps[1], b); upsample(comps[2], c);
        bool is_rgb = adobe_transform == 0 || (adobe_transform < 0 && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
        for (size_t i = 0; i < (size_t)W * H; i++) {
            if (is_rgb) { rgb[3 * i] = a[i]; rgb[3 * i + 1] = b[i]; rgb[3 * i + 2] = c[i]; continue; }
            double y = a[i], cb = (double)b[i] - 128.0, cr = (double)c[i] - 128.0;
            double r = y + 1.402 * cr, g = y - 0.344136286 * cb - 0.714136286 * cr, bl = y + 1.772 * cb;
            auto cl = [](double v) { int iv = (int)std::floor(v + 0.5); return (uint8_t)(iv < 0 ? 0 : iv > 255 ? 255 : iv); };
            rgb[3 * i] = cl(r); rgb[3 * i + 1] = cl(g); rgb[3 * i + 2] = cl(bl);
        }
        return true;
    }
};

} // namespace

int decode_jpeg(const uint8_t* data, size_t size, int* w, int* h, std::vector<uint8_t>& rgb)
{
    Decoder d;
    d.data = data; d.size = size;
    memset(d.qt, 0, sizeof(d.qt));
    if (!d.run(*w, *h, rgb) || d.err) { set_error("JPEG: %s", d.err ? d.err : "decode failed"); return PT_ERR_IO; }
    return PT_OK;
}

} // namespace ptamd
