// Output side beyond the reference's svpng: a compressed PNG (same pixels, smaller file), a linear float
// image (PFM) and a frame checkpoint.  The reference writes one uncompressed 8-bit PNG and never closes it
// (MTPC/MTPC.cpp:10-33, MTPC/svpng.inc); png_writer.cpp reproduces those bytes, this file is what comes after.
// Host-side only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "scene.hpp"

namespace mcpt {
namespace {

// ---------------------------------------------------------------------------------------------- deflate (RFC 1951)
// LZ77 with hash chains over a 32-KiB window, fixed Huffman codes (BTYPE = 01): no code-length tables to build, and on
// path-traced pixels (noise in the low bits) dynamic codes gain little over them.
class BitSink {
public:
    explicit BitSink(std::vector<uint8_t>& o) : out_(o) {}
    void bits(uint32_t v, int n) { acc_ |= uint64_t(v) << fill_; fill_ += n; while (fill_ >= 8) { out_.push_back(uint8_t(acc_)); acc_ >>= 8; fill_ -= 8; } }
    void huff(uint32_t code, int n) { uint32_t r = 0; for (int i = 0; i < n; i++) r |= ((code >> i) & 1u) << (n - 1 - i); bits(r, n); }   // codes go MSB first
    void flush() { if (fill_) { out_.push_back(uint8_t(acc_)); acc_ = 0; fill_ = 0; } }
private:
    std::vector<uint8_t>& out_; uint64_t acc_ = 0; int fill_ = 0;
};

void put_literal(BitSink& b, int sym)           // literal/length alphabet, fixed code (RFC 1951 3.2.6)
{
    if (sym < 144) b.huff(0x30 + sym, 8);
    else if (sym < 256) b.huff(0x190 + (sym - 144), 9);
    else if (sym < 280) b.huff(sym - 256, 7);
    else b.huff(0xC0 + (sym - 280), 8);
}

const int kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const int kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const int kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const int kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

void put_match(BitSink& b, int len, int dist)
{
    int lc = 28;
    while (kLenBase[lc] > len) lc--;
    put_literal(b, 257 + lc);
    if (kLenExtra[lc]) b.bits(uint32_t(len - kLenBase[lc]), kLenExtra[lc]);
    int dc = 29;
    while (kDistBase[dc] > dist) dc--;
    b.huff(uint32_t(dc), 5);
    if (kDistExtra[dc]) b.bits(uint32_t(dist - kDistBase[dc]), kDistExtra[dc]);
}

void deflate_fixed(const std::vector<uint8_t>& in, std::vector<uint8_t>& out)
{
    BitSink b(out);
    b.bits(1, 1); b.bits(1, 2);                  // BFINAL = 1, BTYPE = 01
    const size_t n = in.size();
    const int kHashBits = 15, kWindow = 32768, kMaxChain = 48;
    std::vector<int32_t> head(size_t(1) << kHashBits, -1), prev(n ? n : 1, -1);
    auto hash3 = [&](size_t i) { return ((uint32_t(in[i]) << 10) ^ (uint32_t(in[i + 1]) << 5) ^ uint32_t(in[i + 2])) & ((1u << kHashBits) - 1u); };
    auto insert = [&](size_t i) { if (i + 2 < n) { const uint32_t h = hash3(i); prev[i] = head[h]; head[h] = int32_t(i); } };
    size_t i = 0;
    while (i < n) {
        int best_len = 0, best_dist = 0;
        if (i + 2 < n) {
            int chain = kMaxChain;
            for (int32_t c = head[hash3(i)]; c >= 0 && chain-- > 0 && i - size_t(c) <= size_t(kWindow); c = prev[size_t(c)]) {
                const size_t limit = std::min<size_t>(258, n - i);
                size_t l = 0;
                while (l < limit && in[size_t(c) + l] == in[i + l]) l++;
                if (int(l) > best_len) { best_len = int(l); best_dist = int(i - size_t(c)); if (l == limit) break; }
            }
        }
        if (best_len >= 3) {
            put_match(b, best_len, best_dist);
            for (int k = 0; k < best_len; k++) insert(i + size_t(k));
            i += size_t(best_len);
        } else {
            put_literal(b, in[i]);
            insert(i);
            i++;
        }
    }
    put_literal(b, 256);                         // end of block
    b.flush();
}

// ---------------------------------------------------------------------------------------------- PNG pieces
uint32_t crc32_of(const uint8_t* p, size_t n, uint32_t crc)
{
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t k = 0; k < 256; k++) { uint32_t c = k; for (int j = 0; j < 8; j++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1); table[k] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ p[i]) & 255u] ^ (crc >> 8);
    return crc;
}

void chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& body)
{
    auto be32 = [&](uint32_t u) { out.push_back(uint8_t(u >> 24)); out.push_back(uint8_t(u >> 16)); out.push_back(uint8_t(u >> 8)); out.push_back(uint8_t(u)); };
    be32(uint32_t(body.size()));
    const size_t at = out.size();
    out.insert(out.end(), tag, tag + 4);
    out.insert(out.end(), body.begin(), body.end());
    be32(~crc32_of(out.data() + at, out.size() - at, 0xFFFFFFFFu));
}

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

// 8-bit RGB PNG with a real deflate stream; each scanline takes the filter (None/Sub/Up/Average/Paeth) with the smallest
// sum of absolute residuals.  Decodes to exactly the pixels png_encode() stores.
int64_t png_encode_deflate(const uint8_t* rgb8, int w, int h, uint8_t* out, int64_t cap)
{
    if (!rgb8 || w <= 0 || h <= 0) return -1;
    const size_t pitch = size_t(w) * 3;
    std::vector<uint8_t> raw;
    raw.reserve((pitch + 1) * size_t(h));
    std::vector<uint8_t> cand[5];
    for (auto& c : cand) c.resize(pitch);
    const std::vector<uint8_t> zero(pitch, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = rgb8 + size_t(y) * pitch;
        const uint8_t* up = y ? row - pitch : zero.data();
        long best_sum = -1; int best = 0;
        for (int f = 0; f < 5; f++) {
            long sum = 0;
            for (size_t x = 0; x < pitch; x++) {
                const int a = x >= 3 ? row[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0;
                int pred = 0;
                if (f == 1) pred = a; else if (f == 2) pred = b; else if (f == 3) pred = (a + b) >> 1; else if (f == 4) pred = paeth(a, b, c);
                const uint8_t r = uint8_t(row[x] - pred);
                cand[f][x] = r;
                sum += r < 128 ? r : 256 - r;
            }
            if (best_sum < 0 || sum < best_sum) { best_sum = sum; best = f; }
        }
        raw.push_back(uint8_t(best));
        raw.insert(raw.end(), cand[best].begin(), cand[best].end());
    }
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x9C);        // zlib header: deflate, 32-KiB window, check bits
    deflate_fixed(raw, z);
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) { a = (a + v) % 65521; b = (b + a) % 65521; }
    const uint32_t adler = (b << 16) | a;
    z.push_back(uint8_t(adler >> 24)); z.push_back(uint8_t(adler >> 16)); z.push_back(uint8_t(adler >> 8)); z.push_back(uint8_t(adler));

    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr(13);
    for (int i = 0; i < 4; i++) { ihdr[size_t(i)] = uint8_t(uint32_t(w) >> (24 - 8 * i)); ihdr[size_t(4 + i)] = uint8_t(uint32_t(h) >> (24 - 8 * i)); }
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk(file, "IHDR", ihdr);
    chunk(file, "IDAT", z);
    chunk(file, "IEND", std::vector<uint8_t>());
    if (out && int64_t(file.size()) <= cap) std::memcpy(out, file.data(), file.size());
    else if (out) return -1;
    return int64_t(file.size());
}

// Portable float map: "PF\n<w> <h>\n-1.0\n" + little-endian float32 RGB, bottom row first.  Linear radiance, no clamp.
int write_pfm(const char* file, const double* img, int w, int h, std::string& err)
{
    FILE* fp = std::fopen(file, "wb");
    if (!fp) { err = std::string("cannot open ") + file; return MCPT_ERR_IO; }
    std::fprintf(fp, "PF\n%d %d\n-1.0\n", w, h);
    std::vector<float> row(size_t(w) * 3);
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; y--) {
        const double* src = img + size_t(y) * size_t(w) * 3;
        for (size_t x = 0; x < row.size(); x++) row[x] = float(src[x]);
        ok = std::fwrite(row.data(), sizeof(float), row.size(), fp) == row.size();
    }
    ok = std::fclose(fp) == 0 && ok;
    if (!ok) { err = std::string("short write to ") + file; return MCPT_ERR_IO; }
    return MCPT_OK;
}

// Frame checkpoint: the fp64 frame plus which of `parts` tile partitions are finished.  A frame rendered partition by
// partition is bit-identical to one rendered at once (every (pixel, sample) owns its RNG key), so a resumed frame is too.
namespace {
struct CheckpointHeader {
    char magic[8];               // "MCPTCKP1"
    int32_t width, height, spp, parts;
    uint64_t seed;
    uint64_t scene_tag;          // faces ^ materials ^ lights of the scene, to refuse a checkpoint of something else
};
}  // namespace

int checkpoint_save(const char* file, const double* img, int w, int h, int spp, uint64_t seed, uint64_t scene_tag, int parts,
                    const uint8_t* done, std::string& err)
{
    const std::string tmp = std::string(file) + ".tmp";
    FILE* fp = std::fopen(tmp.c_str(), "wb");
    if (!fp) { err = "cannot open " + tmp; return MCPT_ERR_IO; }
    CheckpointHeader hd{};
    std::memcpy(hd.magic, "MCPTCKP1", 8);
    hd.width = w; hd.height = h; hd.spp = spp; hd.parts = parts; hd.seed = seed; hd.scene_tag = scene_tag;
    const size_t n = size_t(w) * size_t(h) * 3;
    bool ok = std::fwrite(&hd, sizeof hd, 1, fp) == 1 && std::fwrite(done, 1, size_t(parts), fp) == size_t(parts) &&
              std::fwrite(img, sizeof(double), n, fp) == n;
    ok = std::fclose(fp) == 0 && ok;
    if (ok) ok = std::rename(tmp.c_str(), file) == 0;      // a crash while writing leaves the previous checkpoint intact
    if (!ok) { err = std::string("cannot write checkpoint ") + file; return MCPT_ERR_IO; }
    return MCPT_OK;
}

// MCPT_OK: img and done filled; MCPT_ERR_IO: no such file; MCPT_ERR_PARSE: a checkpoint of a different frame
int checkpoint_load(const char* file, double* img, int w, int h, int spp, uint64_t seed, uint64_t scene_tag, int parts, uint8_t* done,
                    std::string& err)
{
    FILE* fp = std::fopen(file, "rb");
    if (!fp) { err = std::string("cannot open ") + file; return MCPT_ERR_IO; }
    CheckpointHeader hd{};
    const size_t n = size_t(w) * size_t(h) * 3;
    int rc = MCPT_OK;
    if (std::fread(&hd, sizeof hd, 1, fp) != 1 || std::memcmp(hd.magic, "MCPTCKP1", 8) != 0 || hd.width != w || hd.height != h ||
        hd.spp != spp || hd.parts != parts || hd.seed != seed || hd.scene_tag != scene_tag) {
        err = std::string(file) + " is a checkpoint of a different frame"; rc = MCPT_ERR_PARSE;
    } else if (std::fread(done, 1, size_t(parts), fp) != size_t(parts) || std::fread(img, sizeof(double), n, fp) != n) {
        err = std::string(file) + " is truncated"; rc = MCPT_ERR_PARSE;
    }
    std::fclose(fp);
    return rc;
}

}  // namespace mcpt
