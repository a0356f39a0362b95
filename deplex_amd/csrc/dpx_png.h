// dpx_png.h — host-side PNG reader for depth images: any PNG -> one 16-bit grey channel, with the conversion rules of the
// reference's loader call stbi_load_16(path, &w, &h, &channels, STBI_grey) (utils/depth_image.cpp:30-40, built with
// STBI_ONLY_PNG): colour -> luma (77 r + 150 g + 29 b) >> 8, alpha dropped, palette expanded, 1/2/4-bit grey scaled to
// 8 bits, and 8-bit results widened to 16 bits as v * 257 — in that order (an 8-bit file is reduced to grey in 8 bits
// first, a 16-bit file in 16 bits).  Written from the PNG specification (chunks, zlib stream, the five scanline filters,
// Adam7); inflation is zlib's.  File IO only — nothing here runs on the GPU.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace dpxpng {

inline uint32_t be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

struct Image {
  int32_t width = 0, height = 0;
  std::vector<uint16_t> grey;  // height * width
};

// Undo the scanline filters of one (sub-)image in place.  `raw` holds rows of 1 filter byte + stride bytes.
inline bool unfilter(unsigned char* raw, size_t rows, size_t stride, size_t bpp) {
  for (size_t r = 0; r < rows; ++r) {
    unsigned char* cur = raw + r * (stride + 1);
    const unsigned char ft = cur[0];
    unsigned char* x = cur + 1;
    const unsigned char* up = r ? cur - stride : nullptr;  // previous row's data (already unfiltered)
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? x[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      x[i] = static_cast<unsigned char>(x[i] + pred);
    }
  }
  return true;
}

// Reads `path`.  Returns an empty string on success, else what went wrong.
inline std::string read_grey16(const char* path, Image& out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return "cannot open file";
  std::vector<unsigned char> file;
  unsigned char buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return "not a PNG file";
  uint32_t W = 0, H = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<unsigned char> idat, plte;
  bool has_trns = false;  // (only matters for the channel count: alpha is dropped)
  size_t pos = 8;
  bool ended = false;
  while (pos + 12 <= file.size() && !ended) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    const unsigned char* data = &file[pos + 8];
    if (len > file.size() || pos + 12 + len > file.size()) return "truncated chunk";
    if (pos == 8 && std::memcmp(type, "IHDR", 4)) return "first chunk is not IHDR";  // (stb_image.h: "first not IHDR")
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return "bad IHDR";
      W = be32(data); H = be32(data + 4);
      depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || interlace > 1) return "unsupported compression / filter / interlace method";
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "tRNS", 4)) {
      has_trns = true;
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      ended = true;
    }
    pos += 12 + len;
  }
  (void)has_trns;
  if (ctype < 0 || W == 0 || H == 0 || W > (1u << 24) || H > (1u << 24)) return "missing or invalid IHDR";
  if (static_cast<unsigned long long>(W) * H > (1ull << 28)) return "image too large";  // (a header can claim anything)
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return "invalid colour type";
  }
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!depth_ok) return "invalid bit depth";
  if (ctype == 3 && plte.size() < 3) return "palette image without PLTE";
  const size_t bits_pp = static_cast<size_t>(channels) * depth;
  const size_t bpp = (bits_pp + 7) / 8;  // filter distance in bytes
  // sub-images: one for a non-interlaced file, seven Adam7 passes otherwise
  struct Pass { uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole = {0, 0, 1, 1};
  size_t total = 0;
  const int npass = interlace ? 7 : 1;
  size_t pw[7], ph[7];
  for (int p = 0; p < npass; ++p) {
    const Pass& ps = interlace ? adam7[p] : whole;
    pw[p] = W > ps.x0 ? (W - ps.x0 + ps.dx - 1) / ps.dx : 0;
    ph[p] = H > ps.y0 ? (H - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (pw[p] && ph[p]) total += ph[p] * (1 + (pw[p] * bits_pp + 7) / 8);
  }
  // A header can claim any size: deflate expands at most 1032 : 1, so more pixels than the IDAT bytes can encode mean a
  // corrupt (or hostile) file — refuse before allocating gigabytes for a 100-byte file.
  if (total > idat.size() * 1032ull + 65536ull) return "corrupt image data";
  // Inflate with the loader's tolerance (utils/depth_image.cpp:30-40 -> stbi_load_16, stb_image.h:4626): the decoded stream
  // may be LONGER than the image needs (extra bytes are ignored) and its Adler-32 trailer is not checked; the output
  // buffer grows with the data instead of being sized from the header.
  std::vector<unsigned char> raw;
  {
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return "corrupt image data";
    zs.next_in = idat.data();
    zs.avail_in = static_cast<uInt>(idat.size());
    unsigned char chunk[65536];
    int zr = Z_OK;
    while (raw.size() < total) {
      zs.next_out = chunk;
      zs.avail_out = sizeof(chunk);
      zr = inflate(&zs, Z_NO_FLUSH);
      const size_t got = sizeof(chunk) - zs.avail_out;
      const size_t take = got < total - raw.size() ? got : total - raw.size();
      raw.insert(raw.end(), chunk, chunk + take);
      if (zr == Z_STREAM_END) break;
      if (zr == Z_DATA_ERROR && raw.size() >= total) break;  // (a bad checksum after complete data: stb does not look at it)
      if (zr != Z_OK || (got == 0 && zs.avail_in == 0)) break;
    }
    inflateEnd(&zs);
    if (raw.size() < total) return "corrupt image data";
  }
  out.width = static_cast<int32_t>(W);
  out.height = static_cast<int32_t>(H);
  out.grey.assign(static_cast<size_t>(W) * H, 0);
  const int maxv = (1 << depth) - 1;
  size_t off = 0;
  for (int p = 0; p < npass; ++p) {
    if (!pw[p] || !ph[p]) continue;
    const Pass& ps = interlace ? adam7[p] : whole;
    const size_t stride = (pw[p] * bits_pp + 7) / 8;
    if (!unfilter(raw.data() + off, ph[p], stride, bpp)) return "invalid filter type";
    for (size_t r = 0; r < ph[p]; ++r) {
      const unsigned char* row = raw.data() + off + r * (stride + 1) + 1;
      for (size_t i = 0; i < pw[p]; ++i) {
        // sample k of pixel i
        auto sample = [&](int k) -> uint32_t {
          const size_t s = i * channels + k;
          if (depth == 16) return (uint32_t(row[2 * s]) << 8) | row[2 * s + 1];
          if (depth == 8) return row[s];
          const size_t bit = s * depth;
          return (row[bit >> 3] >> (8 - depth - (bit & 7))) & maxv;
        };
        uint32_t g;
        if (depth == 16) {
          if (ctype == 0 || ctype == 4) g = sample(0);
          else g = ((sample(0) * 77u) + (sample(1) * 150u) + (29u * sample(2))) >> 8;  // 16-bit luma
          g &= 0xFFFFu;
        } else {
          uint32_t g8;
          if (ctype == 3) {
            const uint32_t idx = sample(0);
            const unsigned char* c = idx * 3 + 2 < plte.size() ? &plte[idx * 3] : nullptr;
            g8 = c ? (((c[0] * 77u) + (c[1] * 150u) + (29u * c[2])) >> 8) & 0xFFu : 0u;
          } else if (ctype == 0 || ctype == 4) {
            g8 = sample(0);
            if (depth < 8) g8 *= (depth == 1 ? 255u : (depth == 2 ? 85u : 17u));  // 1/2/4-bit grey spread over 0..255
          } else {
            g8 = (((sample(0) * 77u) + (sample(1) * 150u) + (29u * sample(2))) >> 8) & 0xFFu;
          }
          g = (g8 << 8) + g8;  // 8 -> 16 bits
        }
        out.grey[(static_cast<size_t>(ps.y0) + r * ps.dy) * W + ps.x0 + i * ps.dx] = static_cast<uint16_t>(g);
      }
    }
    off += ph[p] * (stride + 1);
  }
  return std::string();
}

}  // namespace dpxpng
