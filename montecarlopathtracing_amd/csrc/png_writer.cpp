// Uncompressed 8-bit RGB PNG, byte-for-byte what the reference's svpng emits (MTPC/svpng.inc:77-107):
// IHDR, one IDAT holding a zlib stream (78 01) of stored deflate blocks -- one per scanline, each
// starting with filter byte 0 -- closed by the Adler-32, then IEND.  Width must stay below 21845 so a
// scanline fits one stored block (16-bit length), the same limit the reference has.
#include "scene.hpp"

namespace mcpt {
namespace {

struct Crc32 {
    uint32_t table[256];
    Crc32()
    {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[n] = c;
        }
    }
};
const Crc32 kCrc;

class ChunkWriter {
public:
    ChunkWriter(uint8_t* out, int64_t cap) : out_(out), cap_(cap) {}
    void raw(uint8_t b) { if (pos_ < cap_) out_[pos_] = b; else overflow_ = true; pos_++; }
    void raw32(uint32_t u) { raw(u >> 24); raw((u >> 16) & 255); raw((u >> 8) & 255); raw(u & 255); }
    void open(const char tag[4], uint32_t length) { raw32(length); crc_ = 0xFFFFFFFFu; for (int i = 0; i < 4; i++) byte(uint8_t(tag[i])); }
    void byte(uint8_t b) { raw(b); crc_ = kCrc.table[(crc_ ^ b) & 255u] ^ (crc_ >> 8); }
    void be32(uint32_t u) { byte(u >> 24); byte((u >> 16) & 255); byte((u >> 8) & 255); byte(u & 255); }
    void le16(uint32_t u) { byte(u & 255); byte((u >> 8) & 255); }
    void close() { raw32(~crc_); }
    int64_t size() const { return overflow_ ? -1 : pos_; }
private:
    uint8_t* out_; int64_t cap_; int64_t pos_ = 0; uint32_t crc_ = 0; bool overflow_ = false;
};

}  // namespace

int64_t png_encode(const uint8_t* rgb8, int w, int h, uint8_t* out, int64_t cap)
{
    if (w <= 0 || h <= 0 || int64_t(w) * 3 + 1 > 65535) return -1;
    ChunkWriter cw(out, cap);
    const uint8_t signature[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    for (uint8_t b : signature) cw.raw(b);
    cw.open("IHDR", 13);
    cw.be32(uint32_t(w)); cw.be32(uint32_t(h));
    cw.byte(8); cw.byte(2); cw.byte(0); cw.byte(0); cw.byte(0);   // depth 8, truecolour, deflate, no filter, no interlace
    cw.close();
    const uint32_t pitch = uint32_t(w) * 3 + 1;
    cw.open("IDAT", 2 + uint32_t(h) * (5 + pitch) + 4);
    cw.byte(0x78); cw.byte(0x01);
    uint32_t a = 1, b = 0;                                        // Adler-32 over the raw scanlines
    for (int y = 0; y < h; y++) {
        cw.byte(y == h - 1 ? 1 : 0);                              // BFINAL, BTYPE=00
        cw.le16(pitch); cw.le16(~pitch);
        cw.byte(0); b = (b + a) % 65521;                          // filter type 0 (adds 0 to a)
        const uint8_t* row = rgb8 + size_t(y) * size_t(w) * 3;
        for (uint32_t x = 0; x + 1 < pitch; x++) {
            cw.byte(row[x]);
            a = (a + row[x]) % 65521; b = (b + a) % 65521;
        }
    }
    cw.be32((b << 16) | a);
    cw.close();
    cw.open("IEND", 0);
    cw.close();
    return cw.size();
}

}  // namespace mcpt
