// JPEG reader for map_Kd textures (the reference calls cv::imread, MTPC/sceneManagement.h:137; its shipped texture is a
// progressive 4:4:4 JFIF).  Baseline / extended-sequential / progressive Huffman JPEG, 8-bit, 1 or 3 components,
// restart intervals.  The arithmetic follows the published IJG algorithms that OpenCV's and Pillow's libjpeg use, so that the
// decoded raster is the same 8-bit image: the "slow integer" 13-bit fixed-point IDCT, the 16-bit fixed-point YCbCr->RGB
// tables, and the triangle-filter ("fancy") chroma upsampling for 2:1 horizontally / 2:1 both ways.
// Not supported (returns false): arithmetic coding, 12-bit, lossless, hierarchical, CMYK/YCCK.
#include "jpeg_decoder.hpp"

#include <cstdio>
#include <cstring>

namespace mcpt {
namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int mincode[17], maxcode[18], valptr[17];
    void build()
    {
        int code = 0, k = 0;
        for (int l = 1; l <= 16; l++) {
            valptr[l] = k;
            mincode[l] = code;
            code += bits[l];
            k += bits[l];
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int blocks_w = 0, blocks_h = 0;          // blocks actually coded in a non-interleaved scan
    int stride_blocks = 0, rows_blocks = 0;  // padded to whole MCUs
    std::vector<int16_t> coef;               // [rows_blocks][stride_blocks][64], natural (de-zigzagged) order
    int dc_tbl = 0, ac_tbl = 0, pred = 0;
    std::vector<uint8_t> plane;              // decoded samples, stride_blocks*8 wide
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int nbits = 0;
    int marker = 0;                          // pending marker seen in the entropy stream
    bool fill()
    {
        while (nbits <= 24) {
            int b = 0;
            if (marker == 0 && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int b2 = p < end ? *p : 0xD9;
                    if (b2 == 0) p++;
                    else { marker = b2; p--; b = 0; }   // leave the marker in place, feed zeros
                }
            }
            acc |= uint32_t(b) << (24 - nbits);
            nbits += 8;
        }
        return true;
    }
    int bit() { if (nbits < 1) fill(); const int r = acc >> 31; acc <<= 1; nbits--; return r; }
    int bits(int n)
    {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int r = int(acc >> (32 - n));
        acc <<= n; nbits -= n;
        return r;
    }
    void reset() { acc = 0; nbits = 0; marker = 0; }
};

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

struct Decoder {
    const uint8_t* data; size_t size;
    std::string& err;
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    Component comp[3];
    int restart_interval = 0;
    int adobe_transform = -1;
    BitReader br{nullptr, nullptr};
    int eobrun = 0;

    Decoder(const uint8_t* d, size_t n, std::string& e) : data(d), size(n), err(e) {}
    bool fail(const char* m) { err = m; return false; }

    int decode_symbol(const Huff& h)
    {
        int code = 0;
        for (int l = 1; l <= 16; l++) {
            code = (code << 1) | br.bit();
            if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
        return 0;
    }

    int16_t* block(Component& c, int bx, int by) { return c.coef.data() + (size_t(by) * c.stride_blocks + bx) * 64; }

    // ---- one block of a sequential scan
    void seq_block(Component& c, int16_t* b)
    {
        const int s = decode_symbol(dc[c.dc_tbl]);
        const int diff = s ? extend(br.bits(s), s) : 0;
        c.pred += diff;
        b[0] = int16_t(c.pred);
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(ac[c.ac_tbl]);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
            k += r;
            if (k > 63) break;
            b[kZigzag[k]] = int16_t(extend(br.bits(sz), sz));
            k++;
        }
    }
    // ---- progressive pieces (ITU T.81 G.1.2)
    void dc_first(Component& c, int16_t* b, int al)
    {
        const int s = decode_symbol(dc[c.dc_tbl]);
        const int diff = s ? extend(br.bits(s), s) : 0;
        c.pred += diff;
        b[0] = int16_t(c.pred * (1 << al));
    }
    void dc_refine(int16_t* b, int al) { if (br.bit()) b[0] = int16_t(b[0] | (1 << al)); }
    void ac_first(Component& c, int16_t* b, int ss, int se, int al)
    {
        if (eobrun > 0) { eobrun--; return; }
        for (int k = ss; k <= se; k++) {
            const int rs = decode_symbol(ac[c.ac_tbl]);
            const int r = rs >> 4, s = rs & 15;
            if (s) {
                k += r;
                if (k > 63) return;
                b[kZigzag[k]] = int16_t(extend(br.bits(s), s) * (1 << al));
            } else {
                if (r == 15) k += 15;
                else { eobrun = 1 << r; if (r) eobrun += br.bits(r); eobrun--; break; }
            }
        }
    }
    void ac_refine(Component& c, int16_t* b, int ss, int se, int al)
    {
        const int p1 = 1 << al, m1 = -(1 << al);
        int k = ss;
        if (eobrun == 0) {
            for (; k <= se; k++) {
                const int rs = decode_symbol(ac[c.ac_tbl]);
                int r = rs >> 4, s = rs & 15;
                if (s) s = br.bit() ? p1 : m1;
                else if (r != 15) { eobrun = 1 << r; if (r) eobrun += br.bits(r); break; }
                do {
                    int16_t* co = &b[kZigzag[k]];
                    if (*co != 0) {
                        if (br.bit() && (*co & p1) == 0) *co = int16_t(*co + (*co >= 0 ? p1 : m1));
                    } else if (--r < 0) break;
                    k++;
                } while (k <= se);
                if (s && k <= 63) b[kZigzag[k]] = int16_t(s);
            }
        }
        if (eobrun > 0) {
            for (; k <= se; k++) {
                int16_t* co = &b[kZigzag[k]];
                if (*co != 0 && br.bit() && (*co & p1) == 0) *co = int16_t(*co + (*co >= 0 ? p1 : m1));
            }
            eobrun--;
        }
    }

    bool handle_restart(int& expected)
    {
        br.nbits = 0; br.acc = 0;
        // the marker may already have been met by the bit reader, or still lie ahead
        if (br.marker == 0) {
            while (br.p + 1 < br.end && !(br.p[0] == 0xFF && br.p[1] != 0 && br.p[1] != 0xFF)) br.p++;
            if (br.p + 1 >= br.end) return fail("jpeg: missing restart marker");
            br.marker = br.p[1];
        }
        if (br.marker != 0xD0 + expected) return fail("jpeg: restart marker out of sequence");
        br.p += 2; br.marker = 0;
        expected = (expected + 1) & 7;
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        return true;
    }

    bool scan(const uint8_t* hdr, int len, const uint8_t*& next)
    {
        const int ns = hdr[0];
        if (ns < 1 || ns > ncomp || len < 4 + 2 * ns) return fail("jpeg: bad SOS");
        Component* sc[3];
        for (int i = 0; i < ns; i++) {
            const int id = hdr[1 + 2 * i];
            sc[i] = nullptr;
            for (int c = 0; c < ncomp; c++) if (comp[c].id == id) sc[i] = &comp[c];
            if (!sc[i]) return fail("jpeg: scan names an unknown component");
            sc[i]->dc_tbl = hdr[2 + 2 * i] >> 4; sc[i]->ac_tbl = hdr[2 + 2 * i] & 15;
            if (sc[i]->dc_tbl > 3 || sc[i]->ac_tbl > 3) return fail("jpeg: bad table index");
        }
        const int ss = hdr[1 + 2 * ns], se = hdr[2 + 2 * ns], ah = hdr[3 + 2 * ns] >> 4, al = hdr[3 + 2 * ns] & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && ns != 1)) return fail("jpeg: bad progressive scan parameters");
        }
        for (int i = 0; i < ns; i++) {
            const bool need_dc = !progressive || ss == 0, need_ac = !progressive || ss > 0;
            if (need_dc && !(progressive && ah) && !dc[sc[i]->dc_tbl].present) return fail("jpeg: missing DC Huffman table");
            if (need_ac && !ac[sc[i]->ac_tbl].present) return fail("jpeg: missing AC Huffman table");
        }
        br = BitReader{hdr + len, data + size};
        for (int i = 0; i < ncomp; i++) comp[i].pred = 0;
        eobrun = 0;
        int expected = 0, count = 0;
        auto one = [&](Component& c, int bx, int by) {
            int16_t* b = block(c, bx, by);
            if (!progressive) seq_block(c, b);
            else if (ss == 0) { if (ah == 0) dc_first(c, b, al); else dc_refine(b, al); }
            else { if (ah == 0) ac_first(c, b, ss, se, al); else ac_refine(c, b, ss, se, al); }
        };
        if (ns == 1) {
            Component& c = *sc[0];
            for (int by = 0; by < c.blocks_h; by++)
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    if (restart_interval && count == restart_interval) { if (!handle_restart(expected)) return false; count = 0; }
                    one(c, bx, by);
                    count++;
                }
        } else {
            for (int my = 0; my < mcuy; my++)
                for (int mx = 0; mx < mcux; mx++) {
                    if (restart_interval && count == restart_interval) { if (!handle_restart(expected)) return false; count = 0; }
                    for (int i = 0; i < ns; i++)
                        for (int y = 0; y < sc[i]->v; y++)
                            for (int x = 0; x < sc[i]->h; x++) one(*sc[i], mx * sc[i]->h + x, my * sc[i]->v + y);
                    count++;
                }
        }
        // position after the entropy-coded segment: the next marker
        const uint8_t* p = br.marker ? br.p : br.p;
        while (p + 1 < data + size && !(p[0] == 0xFF && p[1] != 0 && p[1] != 0xFF && !(p[1] >= 0xD0 && p[1] <= 0xD7))) p++;
        next = p;
        return true;
    }

    // ---- IDCT: 13-bit fixed point, two passes, rounding exactly as the IJG "islow" method
    static inline uint8_t range_limit(int x)
    {
        const int v = x & 1023;
        if (v < 128) return uint8_t(128 + v);
        if (v < 512) return 255;
        if (v < 896) return 0;
        return uint8_t(v - 896);
    }
    static void idct(const int16_t* in, const uint16_t* q, uint8_t* out, int stride)
    {
        const int CB = 13, P1 = 2;
        const int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                  F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
        int ws[64];
        auto descale = [](long x, int n) { return int((x + (1L << (n - 1))) >> n); };
        for (int c = 0; c < 8; c++) {
            const int16_t* ip = in + c; const uint16_t* qp = q + c; int* wp = ws + c;
            if (!ip[8] && !ip[16] && !ip[24] && !ip[32] && !ip[40] && !ip[48] && !ip[56]) {
                const int dcv = int(ip[0]) * qp[0] * (1 << P1);
                for (int r = 0; r < 8; r++) wp[8 * r] = dcv;
                continue;
            }
            long z2 = long(ip[16]) * qp[16], z3 = long(ip[48]) * qp[48];
            long z1 = (z2 + z3) * F0541;
            long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            z2 = long(ip[0]) * qp[0]; z3 = long(ip[32]) * qp[32];
            long tmp0 = (z2 + z3) * (1L << CB), tmp1 = (z2 - z3) * (1L << CB);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = long(ip[56]) * qp[56]; tmp1 = long(ip[40]) * qp[40]; tmp2 = long(ip[24]) * qp[24]; tmp3 = long(ip[8]) * qp[8];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            wp[0] = descale(tmp10 + tmp3, CB - P1); wp[56] = descale(tmp10 - tmp3, CB - P1);
            wp[8] = descale(tmp11 + tmp2, CB - P1); wp[48] = descale(tmp11 - tmp2, CB - P1);
            wp[16] = descale(tmp12 + tmp1, CB - P1); wp[40] = descale(tmp12 - tmp1, CB - P1);
            wp[24] = descale(tmp13 + tmp0, CB - P1); wp[32] = descale(tmp13 - tmp0, CB - P1);
        }
        for (int r = 0; r < 8; r++) {
            const int* wp = ws + 8 * r; uint8_t* op = out + size_t(r) * stride;
            if (!wp[1] && !wp[2] && !wp[3] && !wp[4] && !wp[5] && !wp[6] && !wp[7]) {
                const uint8_t v = range_limit(descale(wp[0], P1 + 3));
                for (int c = 0; c < 8; c++) op[c] = v;
                continue;
            }
            long z2 = wp[2], z3 = wp[6];
            long z1 = (z2 + z3) * F0541;
            long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
            long tmp0 = (long(wp[0]) + wp[4]) * (1L << CB), tmp1 = (long(wp[0]) - wp[4]) * (1L << CB);
            const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = wp[7]; tmp1 = wp[5]; tmp2 = wp[3]; tmp3 = wp[1];
            z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; long z4 = tmp1 + tmp3;
            const long z5 = (z3 + z4) * F1175;
            tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
            z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
            z3 += z5; z4 += z5;
            tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
            const int S = CB + P1 + 3;
            op[0] = range_limit(descale(tmp10 + tmp3, S)); op[7] = range_limit(descale(tmp10 - tmp3, S));
            op[1] = range_limit(descale(tmp11 + tmp2, S)); op[6] = range_limit(descale(tmp11 - tmp2, S));
            op[2] = range_limit(descale(tmp12 + tmp1, S)); op[5] = range_limit(descale(tmp12 - tmp1, S));
            op[3] = range_limit(descale(tmp13 + tmp0, S)); op[4] = range_limit(descale(tmp13 - tmp0, S));
        }
    }

    // ---- chroma upsampling to full resolution (component plane -> width x height)
    void upsample(const Component& c, std::vector<uint8_t>& out)
    {
        out.assign(size_t(width) * height, 0);
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;   // real component size
        const int ps = c.stride_blocks * 8;
        const int hx = hmax / c.h, vy = vmax / c.v;
        auto px = [&](int x, int y) -> int { return c.plane[size_t(y) * ps + x]; };
        if (hx == 1 && vy == 1) {
            for (int y = 0; y < height; y++) std::memcpy(&out[size_t(y) * width], &c.plane[size_t(y) * ps], size_t(width));
        } else if (hx == 2 && vy == 1 && hmax % c.h == 0) {                       // h2v1 triangle filter
            for (int y = 0; y < height; y++) {
                std::vector<int> row(size_t(cw) * 2);
                for (int i = 0; i < cw; i++) {
                    const int cur = px(i, y), left = px(i > 0 ? i - 1 : 0, y), right = px(i + 1 < cw ? i + 1 : cw - 1, y);
                    row[2 * i] = i == 0 ? cur : (cur * 3 + left + 1) >> 2;
                    row[2 * i + 1] = i == cw - 1 ? cur : (cur * 3 + right + 2) >> 2;
                }
                for (int x = 0; x < width; x++) out[size_t(y) * width + x] = uint8_t(row[x]);
            }
        } else if (hx == 2 && vy == 2 && hmax % c.h == 0 && vmax % c.v == 0) {   // h2v2 triangle filter
            std::vector<int> sum(static_cast<size_t>(cw), 0);
            for (int oy = 0; oy < height; oy++) {
                const int r = oy >> 1;
                int other = (oy & 1) ? r + 1 : r - 1;
                if (other < 0) other = 0;
                if (other > ch - 1) other = ch - 1;
                const int near = r > ch - 1 ? ch - 1 : r;
                for (int i = 0; i < cw; i++) sum[i] = px(i, near) * 3 + px(i, other);
                for (int i = 0; i < cw; i++) {
                    const int cur = sum[i], left = sum[i > 0 ? i - 1 : 0], right = sum[i + 1 < cw ? i + 1 : cw - 1];
                    const int a = i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + left + 8) >> 4;
                    const int b = i == cw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + right + 7) >> 4;
                    if (2 * i < width) out[size_t(oy) * width + 2 * i] = uint8_t(a);
                    if (2 * i + 1 < width) out[size_t(oy) * width + 2 * i + 1] = uint8_t(b);
                }
            }
        } else {                                                                 // other ratios: replication
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    int sx = x * c.h / hmax, sy = y * c.v / vmax;
                    if (sx > cw - 1) sx = cw - 1;
                    if (sy > ch - 1) sy = ch - 1;
                    out[size_t(y) * width + x] = uint8_t(px(sx, sy));
                }
        }
    }

    bool run(int& w, int& h, std::vector<uint8_t>& bgr)
    {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("jpeg: no SOI");
        const uint8_t* p = data + 2; const uint8_t* end = data + size;
        bool have_frame = false, done = false;
        while (!done && p + 4 <= end) {
            if (p[0] != 0xFF) { p++; continue; }
            const int m = p[1];
            if (m == 0xFF) { p++; continue; }
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            const int len = (p[2] << 8) | p[3];
            if (len < 2 || p + 2 + len > end) return fail("jpeg: truncated segment");
            const uint8_t* s = p + 4; const int n = len - 2;
            if (m == 0xDB) {
                for (int i = 0; i < n;) {
                    const int pq = s[i] >> 4, tq = s[i] & 15; i++;
                    if (tq > 3 || pq > 1 || i + 64 * (pq + 1) > n) return fail("jpeg: bad DQT");
                    for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? uint16_t((s[i] << 8) | s[i + 1]) : s[i]; i += pq + 1; }
                    qt_present[tq] = true;
                }
            } else if (m == 0xC4) {
                for (int i = 0; i < n;) {
                    const int tc = s[i] >> 4, th = s[i] & 15; i++;
                    if (tc > 1 || th > 3 || i + 16 > n) return fail("jpeg: bad DHT");
                    Huff& t = tc ? ac[th] : dc[th];
                    int total = 0;
                    for (int l = 1; l <= 16; l++) { t.bits[l] = s[i + l - 1]; total += t.bits[l]; }
                    i += 16;
                    if (total > 256 || i + total > n) return fail("jpeg: bad DHT");
                    std::memcpy(t.vals, s + i, size_t(total)); i += total;
                    t.present = true; t.build();
                }
            } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_frame) return fail("jpeg: multiple frames");
                if (n < 6 || s[0] != 8) return fail("jpeg: only 8-bit precision is supported");
                height = (s[1] << 8) | s[2]; width = (s[3] << 8) | s[4]; ncomp = s[5];
                if (width <= 0 || height <= 0) return fail("jpeg: bad dimensions");
                if (ncomp != 1 && ncomp != 3) return fail("jpeg: only 1 or 3 components are supported");
                if (n < 6 + 3 * ncomp) return fail("jpeg: bad SOF");
                progressive = m == 0xC2;
                for (int c = 0; c < ncomp; c++) {
                    comp[c].id = s[6 + 3 * c]; comp[c].h = s[7 + 3 * c] >> 4; comp[c].v = s[7 + 3 * c] & 15; comp[c].tq = s[8 + 3 * c];
                    if (comp[c].h < 1 || comp[c].h > 4 || comp[c].v < 1 || comp[c].v > 4 || comp[c].tq > 3) return fail("jpeg: bad component");
                    hmax = comp[c].h > hmax ? comp[c].h : hmax; vmax = comp[c].v > vmax ? comp[c].v : vmax;
                }
                mcux = (width + 8 * hmax - 1) / (8 * hmax); mcuy = (height + 8 * vmax - 1) / (8 * vmax);
                for (int c = 0; c < ncomp; c++) {
                    Component& k = comp[c];
                    const int cw = (width * k.h + hmax - 1) / hmax, ch = (height * k.v + vmax - 1) / vmax;
                    k.blocks_w = (cw + 7) / 8; k.blocks_h = (ch + 7) / 8;
                    k.stride_blocks = mcux * k.h; k.rows_blocks = mcuy * k.v;
                    if (ncomp == 1) { k.stride_blocks = k.blocks_w; k.rows_blocks = k.blocks_h; }
                    k.coef.assign(size_t(k.stride_blocks) * k.rows_blocks * 64, 0);
                }
                if (ncomp == 1) { mcux = comp[0].blocks_w; mcuy = comp[0].blocks_h; comp[0].h = comp[0].v = 1; hmax = vmax = 1; }
                have_frame = true;
            } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
                return fail("jpeg: unsupported coding process (lossless / hierarchical / arithmetic)");
            } else if (m == 0xDD) {
                if (n < 2) return fail("jpeg: bad DRI");
                restart_interval = (s[0] << 8) | s[1];
            } else if (m == 0xEE) {
                if (n >= 12 && std::memcmp(s, "Adobe", 5) == 0) adobe_transform = s[11];
            } else if (m == 0xDA) {
                if (!have_frame) return fail("jpeg: SOS before SOF");
                const uint8_t* next = nullptr;
                if (!scan(s, n, next)) return false;
                p = next;
                continue;
            }
            p += 2 + len;
        }
        if (!have_frame) return fail("jpeg: no frame");
        // coefficients -> samples
        for (int c = 0; c < ncomp; c++) {
            Component& k = comp[c];
            if (!qt_present[k.tq]) return fail("jpeg: missing quantisation table");
            const int ps = k.stride_blocks * 8;
            k.plane.assign(size_t(ps) * k.rows_blocks * 8, 0);
            for (int by = 0; by < k.rows_blocks; by++)
                for (int bx = 0; bx < k.stride_blocks; bx++)
                    idct(block(k, bx, by), qt[k.tq], &k.plane[(size_t(by) * 8) * ps + size_t(bx) * 8], ps);
        }
        w = width; h = height;
        bgr.assign(size_t(width) * height * 3, 0);
        if (ncomp == 1) {
            const int ps = comp[0].stride_blocks * 8;
            for (int y = 0; y < height; y++)
                for (int x = 0; x < width; x++) {
                    const uint8_t g = comp[0].plane[size_t(y) * ps + x];
                    uint8_t* o = &bgr[(size_t(y) * width + x) * 3];
                    o[0] = o[1] = o[2] = g;
                }
            return true;
        }
        std::vector<uint8_t> full[3];
        for (int c = 0; c < 3; c++) upsample(comp[c], full[c]);
        const bool ycc = adobe_transform != 0;      // JFIF, or Adobe transform 1
        auto clamp = [](int v) { return uint8_t(v < 0 ? 0 : (v > 255 ? 255 : v)); };
        for (size_t i = 0; i < size_t(width) * height; i++) {
            int r, g, b;
            if (ycc) {
                const int y = full[0][i], cb = full[1][i] - 128, cr = full[2][i] - 128;
                r = y + ((91881 * cr + 32768) >> 16);                                   // FIX(1.40200)
                g = y + ((-22554 * cb + 32768 + (-46802) * cr) >> 16);                  // FIX(0.34414), FIX(0.71414)
                b = y + ((116130 * cb + 32768) >> 16);                                  // FIX(1.77200)
            } else { r = full[0][i]; g = full[1][i]; b = full[2][i]; }
            bgr[i * 3] = clamp(b); bgr[i * 3 + 1] = clamp(g); bgr[i * 3 + 2] = clamp(r);
        }
        return true;
    }
};

}  // namespace

bool decode_jpeg_memory(const uint8_t* data, size_t size, int& width, int& height, std::vector<uint8_t>& bgr, std::string& err)
{
    Decoder d(data, size, err);
    return d.run(width, height, bgr);
}

bool decode_jpeg_file(const std::string& file, int& width, int& height, std::vector<uint8_t>& bgr, std::string& err)
{
    FILE* fp = std::fopen(file.c_str(), "rb");
    if (!fp) { err = "cannot open " + file; return false; }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(fp);
    return decode_jpeg_memory(buf.data(), buf.size(), width, height, bgr, err);
}

}  // namespace mcpt
