// host_contours.cpp — the contour half of the bubble-cleaning chain, host side.
//
// Replaces, on the small per-bubble crops the GPU kernels of clean.hip produce, the reference's
//   cv2.findContours(RETR_EXTERNAL) -> contourArea / moments filter -> drawContours(FILLED) ->
//   findContours again -> largest -> drawContours(FILLED) + boundingRect
// sequence (reference core/image/cleaning.py:340-386).  Contours are inherently sequential border walks over
// a few thousand pixels per bubble; they stay on the host, in native code, and never see the full page.
//
// Semantics restated from OpenCV: a contour is the outer border of an 8-connected blob, walked pixel centre to
// pixel centre (Suzuki-Abe); area and centroid come from Green's theorem over that lattice polygon (so a
// one-pixel-wide stroke has area 0); FILLED drawing = even-odd interior of all polygons drawn together, plus
// their outlines.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mtx_hip.h"

namespace {

struct Pt { int x, y; };
const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};      // counter-clockwise from east (y grows downwards)
const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

struct Grid {
  const uint8_t* p; int w, h;
  bool on(int x, int y) const { return x >= 0 && y >= 0 && x < w && y < h && p[(size_t)y * w + x] != 0; }
};

// labels 8-connected blobs in raster order; start[i] = first (top-most, then left-most) pixel of blob i
void label_blobs(const Grid& g, std::vector<int>& lab, std::vector<Pt>& start) {
  lab.assign((size_t)g.w * g.h, 0);
  std::vector<Pt> stack;
  for (int y = 0; y < g.h; ++y)
    for (int x = 0; x < g.w; ++x) {
      if (!g.on(x, y) || lab[(size_t)y * g.w + x]) continue;
      const int id = (int)start.size() + 1;
      start.push_back({x, y});
      lab[(size_t)y * g.w + x] = id;
      stack.push_back({x, y});
      while (!stack.empty()) {
        const Pt c = stack.back(); stack.pop_back();
        for (int d = 0; d < 8; ++d) {
          const int nx = c.x + DX[d], ny = c.y + DY[d];
          if (g.on(nx, ny) && !lab[(size_t)ny * g.w + nx]) { lab[(size_t)ny * g.w + nx] = id; stack.push_back({nx, ny}); }
        }
      }
    }
}

int dir_of(int dx, int dy) {
  for (int d = 0; d < 8; ++d) if (DX[d] == dx && DY[d] == dy) return d;
  return 0;
}

// outer border of blob `id` from its start pixel s
void trace_outer(const std::vector<int>& lab, int w, int h, int id, Pt s, std::vector<Pt>& out) {
  auto in = [&](int x, int y) { return x >= 0 && y >= 0 && x < w && y < h && lab[(size_t)y * w + x] == id; };
  out.clear();
  out.push_back(s);
  int first = -1;
  for (int k = 0; k < 8; ++k) {                       // clockwise from the west neighbour
    const int d = (4 - k + 8) % 8;
    if (in(s.x + DX[d], s.y + DY[d])) { first = d; break; }
  }
  if (first < 0) return;                              // single pixel
  const Pt i1 = {s.x + DX[first], s.y + DY[first]};
  Pt prev = i1, cur = s;
  for (;;) {
    const int dp = dir_of(prev.x - cur.x, prev.y - cur.y);
    Pt nxt = cur;
    for (int k = 1; k <= 8; ++k) {                    // counter-clockwise, starting after the previous pixel
      const int d = (dp + k) % 8;
      if (in(cur.x + DX[d], cur.y + DY[d])) { nxt = {cur.x + DX[d], cur.y + DY[d]}; break; }
    }
    if (nxt.x == s.x && nxt.y == s.y && cur.x == i1.x && cur.y == i1.y) break;
    prev = cur; cur = nxt;
    out.push_back(cur);
  }
}

void green_sums(const std::vector<Pt>& c, double& a00, double& a10, double& a01) {
  a00 = a10 = a01 = 0.0;
  const size_t n = c.size();
  double xp = c[n - 1].x, yp = c[n - 1].y;
  for (size_t i = 0; i < n; ++i) {
    const double x = c[i].x, y = c[i].y;
    const double dxy = xp * y - x * yp;
    a00 += dxy; a10 += dxy * (xp + x); a01 += dxy * (yp + y);
    xp = x; yp = y;
  }
}

// pixels on or inside the lattice polygon: everything not 4-connected to the outside of its ring
void fill_inside(const std::vector<Pt>& c, int w, int h, std::vector<uint8_t>& inside) {
  const int W = w + 2, H = h + 2;
  std::vector<uint8_t> m((size_t)W * H, 0);           // 1 = ring, 2 = outside
  for (const Pt& p : c) m[(size_t)(p.y + 1) * W + p.x + 1] = 1;
  std::vector<Pt> stack;
  stack.push_back({0, 0}); m[0] = 2;
  const int dx4[4] = {1, -1, 0, 0}, dy4[4] = {0, 0, 1, -1};
  while (!stack.empty()) {
    const Pt q = stack.back(); stack.pop_back();
    for (int d = 0; d < 4; ++d) {
      const int nx = q.x + dx4[d], ny = q.y + dy4[d];
      if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
      uint8_t& v = m[(size_t)ny * W + nx];
      if (v == 0) { v = 2; stack.push_back({nx, ny}); }
    }
  }
  inside.assign((size_t)w * h, 0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) inside[(size_t)y * w + x] = m[(size_t)(y + 1) * W + x + 1] != 2;
}

struct Blob { std::vector<Pt> poly; double area; };

void external_contours(const uint8_t* img, int w, int h, std::vector<Blob>& blobs) {
  Grid g{img, w, h};
  std::vector<int> lab; std::vector<Pt> start;
  label_blobs(g, lab, start);
  blobs.resize(start.size());
  for (size_t i = 0; i < start.size(); ++i) {
    trace_outer(lab, w, h, (int)i + 1, start[i], blobs[i].poly);
    double a00, a10, a01; green_sums(blobs[i].poly, a00, a10, a01);
    blobs[i].area = std::fabs(a00) * 0.5;
  }
}

}  // namespace

extern "C" {

// float32 distance to the nearest zero pixel under the 5x5 chamfer metric cv2.distanceTransform(src, DIST_L2, 5)
// uses: two raster passes over a 16.16 fixed-point image (weights 1, 1.4, 2.1969; outside the image = far).
// Used on mask crops by the conjoined-bubble partition (reference core/image/detection.py:932-968).
MTX_API int mtx_host_chamfer_l2_5x5(const uint8_t* src, int w, int h, float* dist) {
  if (!src || !dist || w < 1 || h < 1) return MTX_ERR_INVALID;
  const int HV = 65536, DG = 91750, LG = 143976, INIT = 0x1fffffff, B = 2;
  const int W = w + 2 * B;
  std::vector<int> t((size_t)W * (h + 2 * B), INIT);
  auto at = [&](int x, int y) -> int& { return t[(size_t)(y + B) * W + x + B]; };
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      if (src[(size_t)y * w + x] == 0) { at(x, y) = 0; continue; }
      int m = at(x - 1, y - 2) + LG, v;
      v = at(x + 1, y - 2) + LG; m = v < m ? v : m;
      v = at(x - 2, y - 1) + LG; m = v < m ? v : m;
      v = at(x - 1, y - 1) + DG; m = v < m ? v : m;
      v = at(x, y - 1) + HV; m = v < m ? v : m;
      v = at(x + 1, y - 1) + DG; m = v < m ? v : m;
      v = at(x + 2, y - 1) + LG; m = v < m ? v : m;
      v = at(x - 1, y) + HV; m = v < m ? v : m;
      at(x, y) = m;
    }
  for (int y = h - 1; y >= 0; --y)
    for (int x = w - 1; x >= 0; --x) {
      int m = at(x, y), v;
      if (m > HV) {
        v = at(x + 1, y + 2) + LG; m = v < m ? v : m;
        v = at(x - 1, y + 2) + LG; m = v < m ? v : m;
        v = at(x + 2, y + 1) + LG; m = v < m ? v : m;
        v = at(x + 1, y + 1) + DG; m = v < m ? v : m;
        v = at(x, y + 1) + HV; m = v < m ? v : m;
        v = at(x - 1, y + 1) + DG; m = v < m ? v : m;
        v = at(x - 2, y + 1) + LG; m = v < m ? v : m;
        v = at(x + 1, y) + HV; m = v < m ? v : m;
        at(x, y) = m;
      }
      dist[(size_t)y * w + x] = (float)m * (1.0f / 65536.0f);
    }
  return MTX_OK;
}

// thr / eroded: [h][w] crops (0 / 255) at page offset (ox, oy) of a page_w x page_h page.
// final_mask: [h][w] out (0 / 255).  bbox: x, y, w, h of the kept text blob in PAGE coordinates.
// returns the number of text fragments that passed the area + centroid test (0 = nothing to clean), < 0 on error
MTX_API int mtx_host_text_mask(const uint8_t* thr, const uint8_t* eroded, int w, int h, int ox, int oy, int page_w, int page_h,
                               double min_area, uint8_t* final_mask, int* bbox) {
  if (!thr || !eroded || !final_mask || !bbox || w < 1 || h < 1) return MTX_ERR_INVALID;
  std::memset(final_mask, 0, (size_t)w * h);
  std::vector<Blob> blobs;
  external_contours(thr, w, h, blobs);
  std::vector<uint8_t> acc((size_t)w * h, 0), inside;
  int valid = 0;
  for (const Blob& b : blobs) {
    if (!(b.area > min_area)) continue;
    double a00, a10, a01; green_sums(b.poly, a00, a10, a01);
    if (std::fabs(a00) <= 1.1920929e-07) continue;
    const double s2 = a00 > 0 ? 0.5 : -0.5, s6 = a00 > 0 ? 1.0 / 6 : -1.0 / 6;
    const double m00 = a00 * s2, m10 = a10 * s6, m01 = a01 * s6;
    if (m00 == 0) continue;
    const int cx = (int)(m10 / m00), cy = (int)(m01 / m00);          // crop coordinates; truncation like int()
    const int px = cx + ox, py = cy + oy;
    if (px < 0 || py < 0 || px >= page_w || py >= page_h) continue;
    if (cx < 0 || cy < 0 || cx >= w || cy >= h || eroded[(size_t)cy * w + cx] != 255) continue;
    ++valid;
    fill_inside(b.poly, w, h, inside);
    for (size_t i = 0; i < acc.size(); ++i) acc[i] ^= inside[i];      // even-odd across all polygons
    for (const Pt& p : b.poly) acc[(size_t)p.y * w + p.x] |= 2;       // outlines are always drawn
  }
  if (!valid) return 0;
  for (auto& v : acc) v = v ? 255 : 0;
  std::vector<Blob> outer;
  external_contours(acc.data(), w, h, outer);
  if (outer.empty()) return 0;
  // cv2.findContours lists the last-found (bottom-most) contour first; max() keeps the first maximum
  int best = (int)outer.size() - 1;
  for (int i = (int)outer.size() - 2; i >= 0; --i) if (outer[i].area > outer[best].area) best = i;
  fill_inside(outer[best].poly, w, h, inside);
  int x0 = w, y0 = h, x1 = -1, y1 = -1;
  for (const Pt& p : outer[best].poly) { x0 = p.x < x0 ? p.x : x0; y0 = p.y < y0 ? p.y : y0; x1 = p.x > x1 ? p.x : x1; y1 = p.y > y1 ? p.y : y1; }
  for (size_t i = 0; i < inside.size(); ++i) final_mask[i] = inside[i] ? 255 : 0;
  bbox[0] = x0 + ox; bbox[1] = y0 + oy; bbox[2] = x1 - x0 + 1; bbox[3] = y1 - y0 + 1;
  return valid;
}

// Outline of the largest blob of a mask (ultralytics `Masks.xy`: the polygon of one instance = its largest external contour):
// up to `cap` points (x, y) of the outer border walk, pixel centres.  Returns the number of points of the outline (which may exceed
// `cap`: call again with a larger buffer), 0 for an empty mask.
MTX_API int mtx_host_mask_outline(const uint8_t* mask, int w, int h, int* xy, int cap) {
  if (!mask || w < 1 || h < 1 || (cap > 0 && !xy)) return MTX_ERR_INVALID;
  std::vector<Blob> blobs;
  external_contours(mask, w, h, blobs);
  if (blobs.empty()) return 0;
  size_t best = 0;
  for (size_t i = 1; i < blobs.size(); ++i) if (blobs[i].poly.size() > blobs[best].poly.size()) best = i;      // "largest" = most points, as masks2segments
  const std::vector<Pt>& c = blobs[best].poly;
  const int n = (int)c.size();
  for (int i = 0; i < n && i < cap; ++i) { xy[2 * i] = c[i].x; xy[2 * i + 1] = c[i].y; }
  return n;
}

}  // extern "C"
