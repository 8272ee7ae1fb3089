// conv_probe.hip — where a slot of the persistent RCAN conv goes (gfx950): the real kernel source, launched directly.
//   timing of the kernel and of its ablations (1: no MFMA, 3: no halo DMA, 4: no epilogue / stores, 8: the generic per-tile address / bounds
//   paths also on interior tiles); the what-ifs of round 2 (profiles/r02_conv_probe_visit_n.log) are no longer in the kernel source
//   ABL 7: shader-clock stamps at the phase boundaries of every slot, for wave 0 of both groups of workgroups 0 and 97
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/conv_probe.hip -o tools/probes/conv_probe
#include "../../mangatranslator_amd/csrc/conv_c64.hip"
#include <cstdio>
#include <vector>
using namespace mtx;

// VAR 0: ReLU (the probe of rounds 2-5), 1: ReLU + fused channel sums (RCAB conv1), 2: out_scale + residual (RCAB conv2)
template <int ABL, int VAR = 0> static float run(ConvC64Params p, unsigned grid, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  constexpr int ACT = VAR == 2 ? MTX_ACT_NONE : MTX_ACT_RELU;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv3x3_c64_kernel<_Float16, ABL, ACT, VAR == 1, VAR == 2>), dim3(grid), dim3(512), 0, 0, p);
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((conv3x3_c64_kernel<_Float16, ABL, ACT, VAR == 1, VAR == 2>), dim3(grid), dim3(512), 0, 0, p);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 1536, W = argc > 2 ? atoi(argv[2]) : 1024;
  const size_t px = (size_t)H * W;
  std::vector<_Float16> hx(px * 64), hw(64 * 9 * 64);
  unsigned r = 12345;
  for (auto& v : hx) { r = r * 1664525u + 1013904223u; v = (_Float16)(((r >> 9) & 0xffff) / 65536.f - 0.5f); }
  for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (_Float16)((((r >> 9) & 0xffff) / 65536.f - 0.5f) * 0.1f); }
  void *dx, *dw, *dy, *dst; float* db;
  hipMalloc(&dx, px * 128); hipMalloc(&dy, px * 128); hipMalloc(&dw, hw.size() * 2); hipMalloc((void**)&db, 256);
  hipMalloc(&dst, 2 * 2 * 64 * 8 * 8); hipMemset(dst, 0, 2 * 2 * 64 * 8 * 8); hipMemset(db, 0, 256);
  hipMemcpy(dx, hx.data(), px * 128, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  ConvC64Params p{};
  p.x = (const unsigned char*)dx; p.w = (const unsigned char*)dw; p.bias = db; p.res = nullptr; p.y = (unsigned char*)dy; p.chan_sum = (float*)dst;
  p.n = 1; p.h = H; p.w_in = W; p.cin = 64; p.cout = 64; p.ldx = 64; p.ldy = 64; p.ldres = 0; p.act = MTX_ACT_RELU; p.act_param = 0; p.res_scale = 0;
  p.ps = 0; p.res_bcast = 0; p.tiles_x = (W + 15) / 16; p.tiles_y = (H + 15) / 16; p.valid_hw = nullptr; p.y_bytes = (unsigned)(px * 128); p.x_bytes = (unsigned)(px * 128);
  const unsigned grid = c64_grid(1, H, W);
  const double bytes = (double)px * 256 + 73728;
  printf("conv 64->64 %dx%d, grid %u\n", W, H, grid);
  float t;
  t = run<0>(p, grid, 20); printf("ABL 0 (the kernel)            %7.1f us  %6.0f GB/s\n", t, bytes / t / 1e3);
  t = run<8>(p, grid, 20); printf("ABL 8 (generic DMA / epilogue)%7.1f us\n", t);
  t = run<1>(p, grid, 20); printf("ABL 1 (no MFMA)               %7.1f us\n", t);
  t = run<3>(p, grid, 20); printf("ABL 3 (no halo DMA)           %7.1f us\n", t);
  t = run<4>(p, grid, 20); printf("ABL 4 (no epilogue / stores)  %7.1f us\n", t);
  // the RCAB pair's own variants (round 6): conv1 = ReLU + channel sums, conv2 = out_scale * conv + residual
  void *dr, *dsum; float* dsc;
  hipMalloc(&dr, px * 128); hipMemcpy(dr, hx.data(), px * 128, hipMemcpyHostToDevice);
  hipMalloc(&dsum, (size_t)grid * 8 * 64 * 4); hipMalloc((void**)&dsc, 256);
  { std::vector<float> sc(64, 0.5f); hipMemcpy(dsc, sc.data(), 256, hipMemcpyHostToDevice); }
  ConvC64Params p1 = p; p1.chan_sum = (float*)dsum;
  ConvC64Params p2 = p; p2.act = MTX_ACT_NONE; p2.res = (const unsigned char*)dr; p2.ldres = 64; p2.res_scale = 1.f; p2.res_bytes = (unsigned)(px * 128); p2.out_scale = dsc;
  t = run<0, 1>(p1, grid, 20); printf("conv1 form (ReLU + sums)      %7.1f us  %6.0f GB/s\n", t, bytes / t / 1e3);
  t = run<4, 1>(p1, grid, 20); printf("   no epilogue / stores       %7.1f us\n", t);
  t = run<1, 1>(p1, grid, 20); printf("   no MFMA                    %7.1f us\n", t);
  t = run<0, 2>(p2, grid, 20); printf("conv2 form (scale + residual) %7.1f us  %6.0f GB/s\n", t, (bytes + px * 128.0) / t / 1e3);
  t = run<4, 2>(p2, grid, 20); printf("   no epilogue / stores       %7.1f us\n", t);
  t = run<1, 2>(p2, grid, 20); printf("   no MFMA                    %7.1f us\n", t);
  t = run<0>(p, grid, 20); printf("ABL 0 again                   %7.1f us\n", t);
  for (int var = 0; var < 3; ++var) {
  hipMemset(dst, 0, 2 * 2 * 64 * 8 * 8);
  ConvC64Params ps = var == 2 ? p2 : p; ps.chan_sum = (float*)dst;       // the stamps go where the sums would (ABL 7 never flushes sums into it: SUM rows are per wave, the stamp area is separate for var 1 below)
  if (var == 0) t = run<7, 0>(ps, grid, 1); else if (var == 2) t = run<7, 2>(ps, grid, 1); else continue;
  printf("---- stamps, variant %d\n", var);
  std::vector<unsigned long long> st(2 * 2 * 64 * 8);
  hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
  for (int wg = 0; wg < 2; ++wg)
    for (int g = 0; g < 2; ++g) {
      printf("workgroup %d group %d: per slot, shader clocks since the slot began [mfma end | dma issued, epilogue+stores issued, wait done | at barrier, past barrier]\n", wg ? 97 : 0, g);
      for (int s = 0; s < 26; ++s) {
        const unsigned long long* e = &st[((wg * 2 + g) * 64 + s) * 8];
        if (!e[0]) continue;
        auto d = [&](int k) { return e[k] ? (long long)(e[k] - e[0]) : -1LL; };
        printf("  slot %2d  %s  mfma %6lld | dma %6lld epi %6lld wait %6lld | bar %6lld out %6lld\n", s, ((s & 1) == g) ? "MFMA" : "mem ", d(1), d(2), d(3), d(4), d(5), d(6));
      }
    }
  }
  return 0;
}
