// Hardware check of the two mtx_plan_time_ops modes (graph difference vs in-graph wall-clock stamps) on a DiT-like op sequence:
// 8 x (GEMM 8704x9216x3072, joint attention T = 8652, 24 heads, d = 128), plain C ABI, no torch.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/stamp_check.cpp -Lmangatranslator_amd/csrc -lmtx_hip -Wl,-rpath,'$ORIGIN/../mangatranslator_amd/csrc' -o tools/stamp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mtx_hip.h"

#define CK(x) do { int rc_ = (x); if (rc_) { printf("FAIL %s: %s\n", #x, mtx_last_error()); return 1; } } while (0)

int main() {
  CK(mtx_init(0));
  const int64_t T = 8652, H = 24, D = 128, M = 8704, N = 9216, K = 3072;
  void *a, *w, *c, *q, *k, *v, *o, *ws_g, *ws_a;
  hipMalloc(&a, M * K * 2); hipMalloc(&w, N * K * 2); hipMalloc(&c, M * N * 2);
  hipMalloc(&q, T * H * D * 2); hipMalloc(&k, T * H * D * 2); hipMalloc(&v, T * H * D * 2); hipMalloc(&o, T * H * D * 2);
  hipMalloc(&ws_g, MTX_GEMM_WORKSPACE_BYTES); hipMalloc(&ws_a, MTX_ATTN_WORKSPACE_BYTES);
  hipMemset(a, 0x3c, M * K * 2); hipMemset(w, 0x3c, N * K * 2);
  hipMemset(q, 0x3d, T * H * D * 2); hipMemset(k, 0x3d, T * H * D * 2); hipMemset(v, 0x3c, T * H * D * 2);
  std::vector<mtx_op> ops;
  std::vector<int> attn_idx, gemm_idx;
  for (int i = 0; i < 8; ++i) {
    mtx_op g; memset(&g, 0, sizeof g);
    g.kind = MTX_OP_GEMM;
    g.u.gemm.a = a; g.u.gemm.w = w; g.u.gemm.c = c; g.u.gemm.m = M; g.u.gemm.n = N; g.u.gemm.k = K;
    g.u.gemm.lda = K; g.u.gemm.ldw = K; g.u.gemm.ldc = N; g.u.gemm.batch = 1; g.u.gemm.alpha = 1.f;
    g.u.gemm.dtype = MTX_BF16; g.u.gemm.out_dtype = MTX_BF16; g.u.gemm.workspace = ws_g; g.u.gemm.workspace_bytes = MTX_GEMM_WORKSPACE_BYTES;
    gemm_idx.push_back((int)ops.size()); ops.push_back(g);
    mtx_op t; memset(&t, 0, sizeof t);
    t.kind = MTX_OP_ATTN;
    mtx_attn_args& x = t.u.attn;
    x.q = q; x.k = k; x.v = v; x.o = o; x.batch = 1; x.heads = H; x.sq = T; x.sk = T; x.d = D;
    x.q_bs = x.k_bs = x.v_bs = x.o_bs = T * H * D; x.q_ss = x.k_ss = x.v_ss = x.o_ss = H * D; x.q_hs = x.k_hs = x.v_hs = x.o_hs = D;
    x.scale = 0.0883883f; x.dtype = MTX_BF16; x.workspace = ws_a; x.workspace_bytes = MTX_ATTN_WORKSPACE_BYTES; x.flags = MTX_ATTN_Q_PRESCALED;
    attn_idx.push_back((int)ops.size()); ops.push_back(t);
  }
  void* plan = nullptr;
  CK(mtx_plan_create(ops.data(), (int)ops.size(), &plan));
  hipStream_t s; hipStreamCreate(&s);
  float ms = 0.f;
  CK(mtx_plan_time(plan, s, 20, 1, &ms));                         // pre-heat
  CK(mtx_plan_time(plan, s, 10, 1, &ms));
  printf("plan replay %.3f ms (8 gemm + 8 attention)\n", ms);
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 2; ++mode) {
      if (mode) setenv("MTX_TIME_OPS", "stamp", 1); else unsetenv("MTX_TIME_OPS");
      float ta = 0.f, tg = 0.f;
      CK(mtx_plan_time_ops(plan, s, attn_idx.data(), 8, 1, &ta));
      CK(mtx_plan_time_ops(plan, s, attn_idx.data(), 8, 8, &ta));
      CK(mtx_plan_time_ops(plan, s, gemm_idx.data(), 8, 8, &tg));
      printf("%-10s attention %.4f ms/launch   gemm %.4f ms/launch   sum x8 %.3f ms\n", mode ? "stamp" : "difference", ta / 64, tg / 64, (ta + tg) / 8);
    }
  mtx_plan_destroy(plan);
  printf("done\n");
  return 0;
}
