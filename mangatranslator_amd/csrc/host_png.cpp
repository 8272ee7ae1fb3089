// host_png.cpp — the batch harness's PNG writer (include/mtx_hip.h mtx_host_png_encode), native and parallel.
//
// The reference writes every finished page through Pillow + oxipng (core/image/image_utils.py:140-150: `oxipng.optimize_from_memory(png,
// level=2, optimize_alpha=True)`), a Rust optimiser that (a) reduces the colour type where that is lossless, (b) picks a filter per row and
// (c) deflates on several threads.  oxipng is not a dependency of this build, and Pillow's own encoder is one thread of zlib: 0.3 s
// (compress_level 2) to 6 s (optimize=True) for a 2048 x 3072 page, 2-10 s for the 4096 x 6144 result of BASELINE config 5 — slower than the
// GPU produces pages (r03: 0.59 pages/s with I/O against 1.23 without).  This writer does the same three things natively:
//   * lossless reductions: RGBA / LA whose alpha is 255 everywhere lose the alpha channel, RGB(A) with R == G == B everywhere become
//     greyscale (manga pages mostly are) — decoded pixels are unchanged, the decoded MODE follows the file like with oxipng;
//   * per row the filter (None / Sub / Up / Average / Paeth) with the smallest sum of absolute filtered bytes (libpng's heuristic);
//   * the filtered rows are cut into stripes deflated concurrently as raw deflate streams, each primed with the last 32 KB of the stripe
//     before it and ended on a byte boundary (Z_SYNC_FLUSH; the last one Z_FINISH): their concatenation behind one zlib header and in front
//     of the combined Adler-32 is ONE valid zlib stream (the pigz construction), carried by one IDAT chunk per stripe.
// Output bytes are not oxipng's (different deflate implementation); what a reader decodes is identical to the input pixels.
#include <stdint.h>
#include <string.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/mtx_hip.h"

namespace {

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// source pixel fetch with the channel reduction applied: oc output channels out of c stored ones
struct Source {
  const uint8_t* px; int w, h, c, oc;
  // byte b of output pixel x in row y
  inline void row(int y, uint8_t* dst) const {
    const uint8_t* s = px + (size_t)y * w * c;
    if (oc == c) { memcpy(dst, s, (size_t)w * c); return; }
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = s + (size_t)x * c;
      uint8_t* d = dst + (size_t)x * oc;
      if (oc == 1) d[0] = p[0];                                   // grey from RGB(A) / LA
      else if (oc == 2) { d[0] = p[0]; d[1] = p[c - 1]; }         // grey + alpha from RGBA
      else { d[0] = p[0]; d[1] = p[1]; d[2] = p[2]; }             // RGB from RGBA
    }
  }
};

void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

void chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t n) {
  put32(out, (uint32_t)n);
  const size_t at = out.size();
  out.insert(out.end(), type, type + 4);
  if (n) out.insert(out.end(), data, data + n);
  put32(out, (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4)));
}

}  // namespace

extern "C" int64_t mtx_host_png_encode(const uint8_t* pixels, int w, int h, int channels, int level, int threads, int reduce,
                                        uint8_t* out, int64_t out_cap) {
  if (!pixels || w < 1 || h < 1 || channels < 1 || channels > 4) return -1;
  level = std::max(0, std::min(9, level));
  threads = std::max(1, std::min(64, threads));
  // ---- lossless colour-type reduction -------------------------------------------------------------------------------------
  int oc = channels;
  if (reduce) {
    const bool has_alpha = channels == 2 || channels == 4;
    std::atomic<bool> opaque(true), grey(channels >= 3);
    auto scan = [&](int y0, int y1) {
      bool op = true, gr = channels >= 3;
      for (int y = y0; y < y1 && (op || gr); ++y) {
        const uint8_t* s = pixels + (size_t)y * w * channels;
        for (int x = 0; x < w; ++x) {
          const uint8_t* p = s + (size_t)x * channels;
          if (has_alpha && p[channels - 1] != 255) op = false;
          if (channels >= 3 && (p[0] != p[1] || p[1] != p[2])) gr = false;
        }
      }
      if (!op) opaque = false;
      if (!gr) grey = false;
    };
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) ts.emplace_back(scan, (int)((int64_t)h * t / threads), (int)((int64_t)h * (t + 1) / threads));
    for (auto& t : ts) t.join();
    const bool drop_alpha = has_alpha && opaque, to_grey = channels >= 3 && grey;
    oc = (to_grey ? 1 : (channels >= 3 ? 3 : 1)) + ((has_alpha && !drop_alpha) ? 1 : 0);
  }
  const Source src{pixels, w, h, channels, oc};
  const size_t stride = (size_t)w * oc, frow = stride + 1;
  // ---- stripes --------------------------------------------------------------------------------------------------------------
  int stripes = threads;
  const size_t min_stripe = 192 * 1024;                               // a stripe much shorter than the 32 KB window times a few costs ratio
  if ((size_t)h * frow / stripes < min_stripe) stripes = (int)std::max<size_t>(1, (size_t)h * frow / min_stripe);
  stripes = std::min(stripes, h);
  std::vector<int> y0(stripes + 1);
  for (int s = 0; s <= stripes; ++s) y0[s] = (int)((int64_t)h * s / stripes);
  std::vector<uint8_t> filtered((size_t)h * frow);
  // ---- filtering (each stripe needs only the raw row above its first one) -----------------------------------------------------------
  auto filter_rows = [&](int ya, int yb) {
    std::vector<uint8_t> cur(stride), prev(stride, 0), cand(stride);
    if (ya > 0) src.row(ya - 1, prev.data());
    for (int y = ya; y < yb; ++y) {
      src.row(y, cur.data());
      uint8_t* dst = filtered.data() + (size_t)y * frow;
      long best = -1; int best_f = 0;
      for (int f = 0; f < 5; ++f) {
        long sum = 0;
        for (size_t i = 0; i < stride; ++i) {
          const int a = i >= (size_t)oc ? cur[i - oc] : 0, b = prev[i], c2 = i >= (size_t)oc ? prev[i - oc] : 0;
          int v;
          switch (f) {
            case 0: v = cur[i]; break;
            case 1: v = cur[i] - a; break;
            case 2: v = cur[i] - b; break;
            case 3: v = cur[i] - ((a + b) >> 1); break;
            default: v = cur[i] - paeth(a, b, c2); break;
          }
          const uint8_t u = (uint8_t)v;
          cand[i] = u;
          sum += u < 128 ? u : 256 - u;
          if (best >= 0 && sum >= best) break;                       // cannot win any more
        }
        if (best < 0 || sum < best) { best = sum; best_f = f; memcpy(dst + 1, cand.data(), stride); }
        if (level == 0) break;                                        // stored data: filter None
      }
      dst[0] = (uint8_t)best_f;
      prev.swap(cur);
    }
  };
  {
    std::vector<std::thread> ts;
    for (int s = 0; s < stripes; ++s) ts.emplace_back(filter_rows, y0[s], y0[s + 1]);
    for (auto& t : ts) t.join();
  }
  // ---- deflate, one raw stream per stripe -------------------------------------------------------------------------------------------
  std::vector<std::vector<uint8_t>> comp(stripes);
  std::vector<uLong> adler(stripes);
  std::atomic<int> failed(0);
  auto deflate_stripe = [&](int s) {
    const uint8_t* data = filtered.data() + (size_t)y0[s] * frow;
    const size_t n = (size_t)(y0[s + 1] - y0[s]) * frow;
    adler[s] = adler32(adler32(0L, Z_NULL, 0), data, (uInt)n);
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, level == 0 ? Z_DEFAULT_STRATEGY : Z_FILTERED) != Z_OK) { failed = 1; return; }
    if (s > 0) {
      const size_t dict = std::min<size_t>(32768, (size_t)y0[s] * frow);
      deflateSetDictionary(&z, data - dict, (uInt)dict);
    }
    comp[s].resize(deflateBound(&z, (uLong)n) + 16);
    z.next_in = const_cast<Bytef*>(data); z.avail_in = (uInt)n;
    z.next_out = comp[s].data(); z.avail_out = (uInt)comp[s].size();
    const int rc = deflate(&z, s == stripes - 1 ? Z_FINISH : Z_SYNC_FLUSH);
    if ((s == stripes - 1 && rc != Z_STREAM_END) || (s != stripes - 1 && rc != Z_OK) || z.avail_in != 0) failed = 1;
    comp[s].resize(comp[s].size() - z.avail_out);
    deflateEnd(&z);
  };
  {
    std::vector<std::thread> ts;
    for (int s = 0; s < stripes; ++s) ts.emplace_back(deflate_stripe, s);
    for (auto& t : ts) t.join();
  }
  if (failed) return -2;
  uLong ad = adler[0];
  for (int s = 1; s < stripes; ++s) ad = adler32_combine(ad, adler[s], (z_off_t)((size_t)(y0[s + 1] - y0[s]) * frow));
  // ---- the file -------------------------------------------------------------------------------------------------------------------
  std::vector<uint8_t> png;
  size_t total = 0;
  for (auto& c2 : comp) total += c2.size();
  png.reserve(total + 12 * (stripes + 3) + 64);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  png.insert(png.end(), sig, sig + 8);
  uint8_t ihdr[13];
  ihdr[0] = w >> 24; ihdr[1] = w >> 16; ihdr[2] = w >> 8; ihdr[3] = w; ihdr[4] = h >> 24; ihdr[5] = h >> 16; ihdr[6] = h >> 8; ihdr[7] = h;
  ihdr[8] = 8; ihdr[9] = oc == 1 ? 0 : (oc == 2 ? 4 : (oc == 3 ? 2 : 6)); ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk(png, "IHDR", ihdr, 13);
  for (int s = 0; s < stripes; ++s) {
    std::vector<uint8_t> body;
    body.reserve(comp[s].size() + 6);
    if (s == 0) { body.push_back(0x78); body.push_back(level >= 7 ? 0xda : (level >= 6 ? 0x9c : (level >= 2 ? 0x5e : 0x01))); }
    body.insert(body.end(), comp[s].begin(), comp[s].end());
    if (s == stripes - 1) put32(body, (uint32_t)ad);
    chunk(png, "IDAT", body.data(), body.size());
  }
  chunk(png, "IEND", nullptr, 0);
  if (out && (int64_t)png.size() <= out_cap) memcpy(out, png.data(), png.size());
  return (int64_t)png.size();
}
