// png_fuzz — structure-aware fuzzer for the host-side PNG reader (csrc/dpx_png.h); CPU only, run under the sanitizers:
//   g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -std=c++17 tools/png_fuzz.cpp -o /tmp/png_fuzz -lz && /tmp/png_fuzz SEED ITERS
// Builds PNG files with random (valid and invalid) headers, palettes, filter bytes, interlacing, truncations and chunk
// splits around correctly deflated random data and checks that the reader either rejects the file or returns
// width * height samples — never reads or writes out of bounds.
#include "../deplex_amd/csrc/dpx_png.h"
#include <random>
static void put32(std::vector<unsigned char>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void chunk(std::vector<unsigned char>& f, const char* t, const std::vector<unsigned char>& d) {
  put32(f, d.size()); f.insert(f.end(), t, t + 4); f.insert(f.end(), d.begin(), d.end()); put32(f, 0);
}
int main(int argc, char** argv) {
  std::mt19937_64 rng(atoi(argv[1]));
  int iters = atoi(argv[2]);
  size_t ok = 0, bad = 0;
  for (int it = 0; it < iters; ++it) {
    static const int ctypes[] = {0, 2, 3, 4, 6, 0, 2, 3, 4, 6, 1, 5, 7};
    static const int depths[] = {1, 2, 4, 8, 16, 8, 16, 3, 32, 0};
    uint32_t W = 1 + rng() % 40, H = 1 + rng() % 40;
    if (rng() % 50 == 0) W = rng() % 3 ? (1u << 20) : 0;
    if (rng() % 50 == 0) H = rng() % 3 ? (1u << 20) : 0xFFFFFFFFu;
    int ct = ctypes[rng() % 13], dp = depths[rng() % 10], il = rng() % 8 == 0 ? 2 : rng() % 2;
    std::vector<unsigned char> f = {137, 80, 78, 71, 13, 10, 26, 10};
    std::vector<unsigned char> ih; put32(ih, W); put32(ih, H); ih.push_back(dp); ih.push_back(ct); ih.push_back(rng() % 30 == 0); ih.push_back(rng() % 30 == 0); ih.push_back(il);
    if (rng() % 40 == 0) ih.pop_back();
    chunk(f, "IHDR", ih);
    if (ct == 3 || rng() % 4 == 0) { std::vector<unsigned char> p(rng() % 4 ? 3 * (1 + rng() % 256) : rng() % 5); for (auto& b : p) b = rng(); chunk(f, "PLTE", p); }
    if (rng() % 4 == 0) { std::vector<unsigned char> p(rng() % 10); chunk(f, "tRNS", p); }
    // raw data: roughly the right size (so that many cases decode), sometimes short / long
    size_t chans = ct == 2 ? 3 : ct == 4 ? 2 : ct == 6 ? 4 : 1;
    size_t approx = size_t(std::min<uint32_t>(H, 64)) * (1 + (size_t(std::min<uint32_t>(W, 64)) * chans * (dp ? dp : 1) + 7) / 8);
    // exact size computed like a writer would (both interlace modes)
    size_t bits = chans * dp, total = 0;
    if (W <= 64 && H <= 64 && dp) {
      if (il == 1) { static const int x0[7]={0,4,0,2,0,1,0}, y0[7]={0,0,4,0,2,0,1}, dx[7]={8,8,4,4,2,2,1}, dy[7]={8,8,8,4,4,2,2};
        for (int p = 0; p < 7; ++p) { size_t pw = W > (uint32_t)x0[p] ? (W - x0[p] + dx[p] - 1) / dx[p] : 0, ph = H > (uint32_t)y0[p] ? (H - y0[p] + dy[p] - 1) / dy[p] : 0; if (pw && ph) total += ph * (1 + (pw * bits + 7) / 8); } }
      else total = H * (1 + (W * bits + 7) / 8);
    } else total = approx;
    if (rng() % 10 == 0) total = total ? rng() % (total + 10) : 3;
    std::vector<unsigned char> raw(total);
    for (auto& b : raw) b = rng();
    // make most filter bytes valid
    if (W <= 64 && H <= 64 && dp && rng() % 8) {
      // walk rows and set filter bytes 0..4 (occasionally 5)
      size_t off = 0;
      auto setrows = [&](size_t pw, size_t ph) { size_t st = (pw * bits + 7) / 8; for (size_t r = 0; r < ph && off < raw.size(); ++r) { raw[off] = rng() % 60 == 0 ? 5 : rng() % 5; off += st + 1; } };
      if (il == 1) { static const int x0[7]={0,4,0,2,0,1,0}, y0[7]={0,0,4,0,2,0,1}, dx[7]={8,8,4,4,2,2,1}, dy[7]={8,8,8,4,4,2,2};
        for (int p = 0; p < 7; ++p) { size_t pw = W > (uint32_t)x0[p] ? (W - x0[p] + dx[p] - 1) / dx[p] : 0, ph = H > (uint32_t)y0[p] ? (H - y0[p] + dy[p] - 1) / dy[p] : 0; if (pw && ph) setrows(pw, ph); } }
      else setrows(W, H);
    }
    uLongf cl = compressBound(raw.size());
    std::vector<unsigned char> comp(cl);
    compress(comp.data(), &cl, raw.data(), raw.size());
    comp.resize(cl);
    if (rng() % 30 == 0 && !comp.empty()) comp.resize(rng() % comp.size());
    // split into 1-3 IDAT chunks
    size_t cut = comp.size() ? rng() % (comp.size() + 1) : 0;
    chunk(f, "IDAT", std::vector<unsigned char>(comp.begin(), comp.begin() + cut));
    chunk(f, "IDAT", std::vector<unsigned char>(comp.begin() + cut, comp.end()));
    if (rng() % 20) chunk(f, "IEND", {});
    if (rng() % 30 == 0) f.resize(rng() % f.size());
    FILE* o = fopen("/tmp/dpx_png_fuzz_cur.png", "wb"); fwrite(f.data(), 1, f.size(), o); fclose(o);
    dpxpng::Image img;
    std::string e = dpxpng::read_grey16("/tmp/dpx_png_fuzz_cur.png", img);
    if (e.empty()) { ++ok; if (img.grey.size() != size_t(img.width) * img.height) { printf("SIZE MISMATCH\n"); return 1; } } else ++bad;
  }
  printf("ok %zu rejected %zu\n", ok, bad);
}
