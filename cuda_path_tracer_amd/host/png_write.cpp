#include "png_write.h"

#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>
#include <vector>

namespace cpt {
namespace {

void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

bool write_png(FILE* f, int w, int h, const uint8_t* rgb) {
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (1 + (size_t)w * 3));
    for (int row = h - 1; row >= 0; --row) {  // top row of the file = last row of the buffer
        raw.push_back(0);                     // filter type "none"
        raw.insert(raw.end(), rgb + (size_t)row * w * 3, rgb + (size_t)(row + 1) * w * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(clen);
    if (compress2(z.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(clen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    std::vector<uint8_t> ihdr;
    be32(ihdr, (uint32_t)w);
    be32(ihdr, (uint32_t)h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, truecolour, deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    return fwrite(out.data(), 1, out.size(), f) == out.size();
}

bool write_bmp(FILE* f, int w, int h, const uint8_t* rgb) {
    const uint32_t row = ((uint32_t)w * 3 + 3) & ~3u, size = 54 + row * (uint32_t)h;
    uint8_t hdr[54] = {0};
    auto le32 = [&](int at, uint32_t x) { hdr[at] = (uint8_t)x; hdr[at + 1] = (uint8_t)(x >> 8); hdr[at + 2] = (uint8_t)(x >> 16); hdr[at + 3] = (uint8_t)(x >> 24); };
    hdr[0] = 'B'; hdr[1] = 'M';
    le32(2, size); le32(10, 54); le32(14, 40); le32(18, (uint32_t)w); le32(22, (uint32_t)h);
    hdr[26] = 1; hdr[28] = 24;
    le32(34, row * (uint32_t)h);
    if (fwrite(hdr, 1, 54, f) != 54) return false;
    std::vector<uint8_t> line(row, 0);
    for (int j = 0; j < h; ++j) {
        for (int i = 0; i < w; ++i) {
            const uint8_t* p = rgb + ((size_t)j * w + i) * 3;
            line[3 * i] = p[2]; line[3 * i + 1] = p[1]; line[3 * i + 2] = p[0];
        }
        if (fwrite(line.data(), 1, row, f) != row) return false;
    }
    return true;
}

}  // namespace

bool write_image(const std::string& path, int w, int h, const uint8_t* rgb, std::string& err) {
    if (w < 1 || h < 1 || !rgb) { err = "write_image: empty image"; return false; }
    std::string ext = path.size() >= 4 ? path.substr(path.size() - 4) : std::string();
    std::transform(ext.begin(), ext.end(), ext.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    if (ext != ".png" && ext != ".bmp") { err = "unknown image type (use .png or .bmp): " + path; return false; }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot open " + path + " for writing"; return false; }
    bool ok = ext == ".png" ? write_png(f, w, h, rgb) : write_bmp(f, w, h, rgb);
    if (fclose(f) != 0) ok = false;
    if (!ok) err = "error while writing " + path;
    return ok;
}

}  // namespace cpt
