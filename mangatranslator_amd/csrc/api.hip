// api.hip — the extern "C" surface of libmtx_hip.so (include/mtx_hip.h) and the plan executor.
//
// A plan is the reference's "model object" seen from below: the Python host mirrors
// ModelManager.load_* (core/ml/model_manager.py:617-1337), builds the op list of one network for
// one input shape, and from then on a forward pass is ONE native call that launches the whole
// static graph on the caller's stream (optionally replayed as a hipGraph).
#include "mtx_device.h"
#include <string>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <new>

namespace mtx {
int conv2d_launch(const mtx_conv2d_args*, void*, const char**);
int conv2d_tiles(const mtx_conv2d_args*);
int gemm_launch(const mtx_gemm_args*, void*, const char**);
void gemm_last_split(int*);
int attn_launch(const mtx_attn_args*, void*, const char**);
int norm_launch(const mtx_norm_args*, void*, const char**);
int groupnorm_launch(const mtx_groupnorm_args*, void*, const char**);
int ew_launch(const mtx_ew_args*, void*, const char**);
int ca_launch(const mtx_ca_args*, void*, const char**);
int img_launch(const mtx_img_args*, void*, const char**);
int resize_thresh_launch(const mtx_resize_thresh_args*, void*, const char**);
int mask_select_launch(const mtx_mask_select_args*, void*, const char**);
int preproc_launch(const mtx_preproc_args*, void*, const char**);
int yolo_decode_launch(const mtx_yolo_decode_args*, void*, const char**);
int clean_launch(const mtx_clean_args*, void*, const char**);
int detr_launch(const mtx_detr_args*, void*, const char**);
int quant_launch(const mtx_quant_args*, void*, const char**);
int tail_launch(const mtx_tail_args*, void*, const char**);

static thread_local std::string g_err;

static int fail(int code, const char* what) {
  g_err = what ? what : "unknown error";
  return code;
}

static int check_launch(int rc, const char* err) {
  if (rc != MTX_OK) return fail(rc, err);
#ifndef MTX_EMU
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_err = std::string("HIP launch failed: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
#endif
  return MTX_OK;
}

struct Plan {
  std::vector<mtx_op> ops;
#ifndef MTX_EMU
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  hipStream_t side = nullptr;              // MTX_LANE_SIDE ops (created on first use)
  std::vector<hipEvent_t> events;          // one per fork / join point of the op list
#endif
};

static int run_op(const mtx_op& op, void* stream) {
  const char* err = nullptr;
  int rc;
  switch (op.kind) {
    case MTX_OP_CONV2D: rc = conv2d_launch(&op.u.conv, stream, &err); break;
    case MTX_OP_GEMM: rc = gemm_launch(&op.u.gemm, stream, &err); break;
    case MTX_OP_ATTN: rc = attn_launch(&op.u.attn, stream, &err); break;
    case MTX_OP_NORM: rc = norm_launch(&op.u.norm, stream, &err); break;
    case MTX_OP_GROUPNORM: rc = groupnorm_launch(&op.u.gn, stream, &err); break;
    case MTX_OP_EW: rc = ew_launch(&op.u.ew, stream, &err); break;
    case MTX_OP_CA: rc = ca_launch(&op.u.ca, stream, &err); break;
    case MTX_OP_IMG: rc = img_launch(&op.u.img, stream, &err); break;
    case MTX_OP_RESIZE_THRESH: rc = resize_thresh_launch(&op.u.rt, stream, &err); break;
    case MTX_OP_MASK_SELECT: rc = mask_select_launch(&op.u.sel, stream, &err); break;
    case MTX_OP_PREPROC: rc = preproc_launch(&op.u.pre, stream, &err); break;
    case MTX_OP_YOLO_DECODE: rc = yolo_decode_launch(&op.u.yd, stream, &err); break;
    case MTX_OP_DETR: rc = detr_launch(&op.u.detr, stream, &err); break;
    case MTX_OP_QUANT: rc = quant_launch(&op.u.quant, stream, &err); break;
    case MTX_OP_TAIL: rc = tail_launch(&op.u.tail, stream, &err); break;
    case MTX_OP_MEMSET:
      if (op.u.ms.bytes < 0 || (!op.u.ms.ptr && op.u.ms.bytes > 0)) { rc = MTX_ERR_INVALID; err = "memset: null pointer or negative size"; break; }
      fill_bytes_async(op.u.ms.ptr, op.u.ms.value, (size_t)op.u.ms.bytes, stream);        // a kernel: a captured hipMemsetAsync (memset node) did not always clear (mtx_device.h)
      rc = MTX_OK;
      break;
    default: rc = MTX_ERR_INVALID; err = "unknown op kind"; break;
  }
  return check_launch(rc, err);
}

#ifndef MTX_EMU
// Opt-in alternative to the graph-difference timing below (MTX_TIME_OPS=stamp): ONE replay graph of the whole plan with a one-lane
// kernel before and after every selected op that stores the device's constant-rate wall clock; the op's time is the stamp
// difference minus two dispatch gaps, the gap taken from a back-to-back stamp pair in the same graph.  Unlike "graph with minus graph without" it does not change the power
// mix of the replay, so the clocks the other kernels run at do not leak into the figure.
__global__ void stamp_kernel(unsigned long long* out) { *out = wall_clock64(); }

static int time_ops_stamped(Plan* p, const std::vector<char>& sel, int iters, float* ms_total) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
    return fail(MTX_ERR_HIP, "mtx_plan_time_ops: wall clock rate unavailable");
  size_t n_sel = 0;
  for (char c : sel) n_sel += c != 0;
  hipStream_t s = nullptr;
  unsigned long long* d_t = nullptr;
  if (hipDeviceSynchronize() != hipSuccess || hipStreamCreate(&s) != hipSuccess) return fail(MTX_ERR_HIP, "mtx_plan_time_ops: stream setup failed");
  if (hipMalloc((void**)&d_t, (2 * n_sel + 2) * sizeof(unsigned long long)) != hipSuccess) { hipStreamDestroy(s); return fail(MTX_ERR_HIP, "mtx_plan_time_ops: hipMalloc failed"); }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int rc = MTX_OK;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: begin capture failed");
  if (rc == MTX_OK) {
    size_t k = 0;
    for (size_t i = 0; i < p->ops.size() && rc == MTX_OK; ++i) {
      if (sel[i] && k == 0) {      // calibration pair: two stamps back to back = one dispatch gap + one stamp kernel
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, d_t + 2 * n_sel);
        hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, d_t + 2 * n_sel + 1);
      }
      if (sel[i]) hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, d_t + 2 * k);
      rc = run_op(p->ops[i], (void*)s);
      if (rc != MTX_OK) g_err = "op " + std::to_string(i) + ": " + g_err;
      if (sel[i]) { hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, s, d_t + 2 * k + 1); ++k; }
    }
    hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc == MTX_OK && e != hipSuccess) rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: end capture failed");
    if (rc == MTX_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: instantiate failed");
  }
  std::vector<unsigned long long> h_t(2 * n_sel + 2);
  double ticks = 0.0;
  for (int it = -1; it < iters && rc == MTX_OK; ++it) {                // replay -1 is untimed
    if (hipGraphLaunch(exec, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: graph launch failed"); break; }
    if (it < 0) continue;
    if (hipMemcpy(h_t.data(), d_t, h_t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: stamp readback failed"); break; }
    // a bracket holds the op plus two dispatch gaps (stamp -> op, op -> stamp); the calibration pair measures one such gap
    const double gap = (double)(h_t[2 * n_sel + 1] - h_t[2 * n_sel]);
    for (size_t k = 0; k < n_sel; ++k) {
      const double d = (double)(h_t[2 * k + 1] - h_t[2 * k]) - 2.0 * gap;
      ticks += d > 0.0 ? d : 0.0;
    }
  }
  if (exec) hipGraphExecDestroy(exec);
  if (graph) hipGraphDestroy(graph);
  hipFree(d_t);
  hipStreamDestroy(s);
  *ms_total = (float)(ticks / (double)khz);
  return rc;
}
#endif

}  // namespace mtx

using namespace mtx;

extern "C" {

int mtx_abi_version(void) { return MTX_ABI_VERSION; }

size_t mtx_abi_sizeof(int kind) {
  switch (kind) {
    case 0: return sizeof(mtx_op);
    case MTX_OP_CONV2D: return sizeof(mtx_conv2d_args);
    case MTX_OP_GEMM: return sizeof(mtx_gemm_args);
    case MTX_OP_ATTN: return sizeof(mtx_attn_args);
    case MTX_OP_NORM: return sizeof(mtx_norm_args);
    case MTX_OP_GROUPNORM: return sizeof(mtx_groupnorm_args);
    case MTX_OP_EW: return sizeof(mtx_ew_args);
    case MTX_OP_CA: return sizeof(mtx_ca_args);
    case MTX_OP_IMG: return sizeof(mtx_img_args);
    case MTX_OP_RESIZE_THRESH: return sizeof(mtx_resize_thresh_args);
    case MTX_OP_MEMSET: return sizeof(mtx_memset_args);
    case MTX_OP_MASK_SELECT: return sizeof(mtx_mask_select_args);
    case MTX_OP_PREPROC: return sizeof(mtx_preproc_args);
    case MTX_OP_YOLO_DECODE: return sizeof(mtx_yolo_decode_args);
    case MTX_OP_DETR: return sizeof(mtx_detr_args);
    case MTX_OP_QUANT: return sizeof(mtx_quant_args);
    case MTX_OP_TAIL: return sizeof(mtx_tail_args);
    case 100: return sizeof(mtx_clean_args);       /* op-level only (not a plan op) */
    default: return 0;
  }
}

const char* mtx_last_error(void) { return g_err.c_str(); }

int mtx_init(int device_ordinal) {
#ifndef MTX_EMU
  hipError_t e = hipSetDevice(device_ordinal);
  if (e != hipSuccess) { g_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device_ordinal);
  if (e != hipSuccess) { g_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_err = std::string("libmtx_hip is built for gfx950 only; device is ") + prop.gcnArchName;
    return MTX_ERR_UNSUPPORTED;
  }
#else
  (void)device_ordinal;
#endif
  return MTX_OK;
}

int mtx_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
#ifndef MTX_EMU
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MTX_ERR_HIP, "hipGetDevice failed");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(MTX_ERR_HIP, "hipGetDeviceProperties failed");
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
#else
  if (cu_count) *cu_count = 0;
  if (lds_bytes) *lds_bytes = 0;
  if (arch && arch_len > 0) { strncpy(arch, "emu", (size_t)arch_len - 1); arch[arch_len - 1] = 0; }
#endif
  return MTX_OK;
}

#define MTX_OP_ENTRY(name, type, fn)                                     \
  int name(const type* a, void* stream) {                                \
    if (!a) return fail(MTX_ERR_INVALID, #name ": null args");           \
    const char* err = nullptr;                                           \
    return check_launch(fn(a, stream, &err), err);                       \
  }
MTX_OP_ENTRY(mtx_conv2d, mtx_conv2d_args, conv2d_launch)
MTX_OP_ENTRY(mtx_gemm, mtx_gemm_args, gemm_launch)
MTX_OP_ENTRY(mtx_attention, mtx_attn_args, attn_launch)
MTX_OP_ENTRY(mtx_norm, mtx_norm_args, norm_launch)
MTX_OP_ENTRY(mtx_groupnorm, mtx_groupnorm_args, groupnorm_launch)
MTX_OP_ENTRY(mtx_elementwise, mtx_ew_args, ew_launch)
MTX_OP_ENTRY(mtx_channel_attention, mtx_ca_args, ca_launch)
MTX_OP_ENTRY(mtx_image_convert, mtx_img_args, img_launch)
MTX_OP_ENTRY(mtx_resize_threshold, mtx_resize_thresh_args, resize_thresh_launch)
MTX_OP_ENTRY(mtx_mask_select, mtx_mask_select_args, mask_select_launch)
MTX_OP_ENTRY(mtx_preprocess, mtx_preproc_args, preproc_launch)
MTX_OP_ENTRY(mtx_yolo_decode, mtx_yolo_decode_args, yolo_decode_launch)
MTX_OP_ENTRY(mtx_bubble_clean, mtx_clean_args, clean_launch)
MTX_OP_ENTRY(mtx_detr, mtx_detr_args, detr_launch)
MTX_OP_ENTRY(mtx_quantize_mx, mtx_quant_args, quant_launch)
MTX_OP_ENTRY(mtx_page_tail, mtx_tail_args, tail_launch)


int mtx_gemm_last_split(int* whole_tiles, int* k_slices, int* tail_pieces) {
  int v[3];
  gemm_last_split(v);
  if (whole_tiles) *whole_tiles = v[0];
  if (k_slices) *k_slices = v[1];
  if (tail_pieces) *tail_pieces = v[2];
  return MTX_OK;
}

int mtx_conv2d_tiles(const mtx_conv2d_args* a) {
  if (!a) return fail(MTX_ERR_INVALID, "mtx_conv2d_tiles: null args");
  int t = conv2d_tiles(a);
  if (t < 0) return fail(MTX_ERR_UNSUPPORTED, "mtx_conv2d_tiles: unsupported kernel/stride");
  return t;
}

int mtx_plan_create(const mtx_op* ops, int n_ops, void** plan) {
  if (!ops || n_ops < 0 || !plan) return fail(MTX_ERR_INVALID, "mtx_plan_create: bad arguments");
  Plan* p = new (std::nothrow) Plan();
  if (!p) return fail(MTX_ERR_STATE, "mtx_plan_create: out of memory");
  p->ops.assign(ops, ops + n_ops);
  *plan = p;
  return MTX_OK;
}

int mtx_plan_num_ops(void* plan) { return plan ? (int)static_cast<Plan*>(plan)->ops.size() : MTX_ERR_INVALID; }

int mtx_plan_run_range(void* plan, int first, int last, void* stream) {
  if (!plan) return fail(MTX_ERR_INVALID, "mtx_plan_run_range: null plan");
  Plan* p = static_cast<Plan*>(plan);
  if (first < 0) first = 0;
  if (last >= (int)p->ops.size()) last = (int)p->ops.size() - 1;
  for (int i = first; i <= last; ++i) {
    int rc = run_op(p->ops[(size_t)i], stream);
    if (rc != MTX_OK) { g_err = "op " + std::to_string(i) + ": " + g_err; return rc; }
  }
  return MTX_OK;
}

// Whole plan with its lanes: side runs fork from the main stream at their first op and join at the next op that carries
// MTX_LANE_JOIN (or at the end).  The same code records the fork / join edges when the main stream is being captured into a hipGraph
// (event record + stream wait inside a capture become graph dependencies), so a replayed plan keeps the two branches parallel.
int mtx_plan_run(void* plan, void* stream) {
  if (!plan) return fail(MTX_ERR_INVALID, "mtx_plan_run: null plan");
  Plan* p = static_cast<Plan*>(plan);
  const int n = (int)p->ops.size();
#ifdef MTX_EMU
  return mtx_plan_run_range(plan, 0, n - 1, stream);
#else
  bool lanes = false;
  for (const mtx_op& op : p->ops) lanes = lanes || (op.lane & MTX_LANE_SIDE);
  if (!lanes || stream == nullptr) return mtx_plan_run_range(plan, 0, n - 1, stream);     // the legacy default stream cannot fork
  hipStream_t main_s = (hipStream_t)stream;
  if (!p->side && hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess) return fail(MTX_ERR_HIP, "mtx_plan_run: side stream");
  size_t ev = 0;
  auto next_event = [&]() -> hipEvent_t {
    if (ev == p->events.size()) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
      p->events.push_back(e);
    }
    return p->events[ev++];
  };
  auto edge = [&](hipStream_t from, hipStream_t to) -> bool {       // `to` continues after everything issued on `from` so far
    hipEvent_t e = next_event();
    return e && hipEventRecord(e, from) == hipSuccess && hipStreamWaitEvent(to, e, 0) == hipSuccess;
  };
  bool side_open = false, prev_side = false;
  for (int i = 0; i < n; ++i) {
    const mtx_op& op = p->ops[(size_t)i];
    const bool on_side = (op.lane & MTX_LANE_SIDE) != 0;
    if (!on_side && (op.lane & MTX_LANE_JOIN) && side_open) {
      if (!edge(p->side, main_s)) return fail(MTX_ERR_HIP, "mtx_plan_run: join failed");
      side_open = false;
    }
    if (on_side && !prev_side) {                                     // a side run begins: it sees every main op recorded before it
      if (!edge(main_s, p->side)) return fail(MTX_ERR_HIP, "mtx_plan_run: fork failed");
      side_open = true;
    }
    int rc = run_op(op, on_side ? (void*)p->side : stream);
    if (rc != MTX_OK) { g_err = "op " + std::to_string(i) + ": " + g_err; if (side_open) edge(p->side, main_s); return rc; }
    prev_side = on_side;
  }
  if (side_open && !edge(p->side, main_s)) return fail(MTX_ERR_HIP, "mtx_plan_run: final join failed");
  return MTX_OK;
#endif
}

int mtx_plan_run_graph(void* plan, void* stream) {
  if (!plan) return fail(MTX_ERR_INVALID, "mtx_plan_run_graph: null plan");
#ifdef MTX_EMU
  return mtx_plan_run(plan, stream);
#else
  Plan* p = static_cast<Plan*>(plan);
  hipStream_t s = (hipStream_t)stream;
  if (!p->exec) {
    if (s == nullptr) return fail(MTX_ERR_INVALID, "mtx_plan_run_graph: capture needs a non-default stream");
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { g_err = std::string("hipStreamBeginCapture: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
    int rc = mtx_plan_run(plan, stream);
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(s, &g);
    if (rc != MTX_OK) { if (g) hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { g_err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
    e = hipGraphInstantiate(&p->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(g); g_err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
    p->graph = g;
  }
  hipError_t e = hipGraphLaunch(p->exec, s);
  if (e != hipSuccess) { g_err = std::string("hipGraphLaunch: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
  return MTX_OK;
#endif
}

void mtx_plan_destroy(void* plan) {
  if (!plan) return;
  Plan* p = static_cast<Plan*>(plan);
#ifndef MTX_EMU
  if (p->exec) hipGraphExecDestroy(p->exec);
  if (p->graph) hipGraphDestroy(p->graph);
  for (hipEvent_t e : p->events) hipEventDestroy(e);
  if (p->side) hipStreamDestroy(p->side);
#endif
  delete p;
}

int mtx_plan_time_range(void* plan, int first, int last, void* stream, int iters, float* ms) {
  if (!plan || !ms || iters < 1) return fail(MTX_ERR_INVALID, "mtx_plan_time_range: bad arguments");
#ifdef MTX_EMU
  for (int i = 0; i < iters; ++i) { int rc = mtx_plan_run_range(plan, first, last, stream); if (rc) return rc; }
  *ms = 0.f;
  return MTX_OK;
#else
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MTX_ERR_HIP, "hipEventCreate failed");
  int rc = MTX_OK;
  hipEventRecord(e0, s);
  for (int i = 0; i < iters && rc == MTX_OK; ++i) rc = mtx_plan_run_range(plan, first, last, stream);
  hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  float t = 0.f;
  if (rc == MTX_OK && e == hipSuccess) { hipEventElapsedTime(&t, e0, e1); *ms = t / (float)iters; }
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc != MTX_OK) return rc;
  if (e != hipSuccess) { g_err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
  return MTX_OK;
#endif
}

// In-context op timing: the whole plan runs eagerly `iters` times with a HIP-event pair around every selected op, so an
// op is measured with the caches, clocks and neighbours it has in the real sequence (isolated back-to-back replays of one
// attention launch measured 13 % faster than the same launch inside a denoising step).  ms_total = sum over ops and iters.
int mtx_plan_time_ops(void* plan, void* stream, const int* op_idx, int n_idx, int iters, float* ms_total) {
  if (!plan || !op_idx || !ms_total || n_idx < 1 || iters < 1) return fail(MTX_ERR_INVALID, "mtx_plan_time_ops: bad arguments");
  Plan* p = static_cast<Plan*>(plan);
  *ms_total = 0.f;
#ifdef MTX_EMU
  for (int i = 0; i < iters; ++i) { int rc = mtx_plan_run(plan, stream); if (rc) return rc; }
  return MTX_OK;
#else
  (void)stream;
  std::vector<char> sel(p->ops.size(), 0);
  for (int i = 0; i < n_idx; ++i) { if (op_idx[i] < 0 || op_idx[i] >= (int)p->ops.size()) return fail(MTX_ERR_INVALID, "mtx_plan_time_ops: op index out of range"); sel[(size_t)op_idx[i]] = 1; }
  const char* how = getenv("MTX_TIME_OPS");
  if (how && !strcmp(how, "stamp")) return time_ops_stamped(p, sel, iters, ms_total);
  // In-context time of the selected ops = replay of the WHOLE plan as a hipGraph minus replay of the same graph without them, on a
  // stream of its own.  (Event pairs around single eager launches read 0.15 ms too long on some boxes — the pair's own cost and the
  // submission gap land inside the interval — while the production path is a graph replay anyway.)
  hipStream_t s = nullptr;
  if (hipDeviceSynchronize() != hipSuccess || hipStreamCreate(&s) != hipSuccess) return fail(MTX_ERR_HIP, "mtx_plan_time_ops: stream setup failed");
  hipGraphExec_t exec[2] = {nullptr, nullptr};
  hipGraph_t graph[2] = {nullptr, nullptr};
  int rc = MTX_OK;
  for (int g = 0; g < 2 && rc == MTX_OK; ++g) {
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: begin capture failed"); break; }
    for (size_t i = 0; i < p->ops.size() && rc == MTX_OK; ++i) {
      if (g == 1 && sel[i]) continue;
      rc = run_op(p->ops[i], (void*)s);
      if (rc != MTX_OK) g_err = "op " + std::to_string(i) + ": " + g_err;
    }
    hipError_t e = hipStreamEndCapture(s, &graph[g]);
    if (rc == MTX_OK && e != hipSuccess) rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: end capture failed");
    if (rc == MTX_OK && hipGraphInstantiate(&exec[g], graph[g], nullptr, nullptr, 0) != hipSuccess) rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: instantiate failed");
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == MTX_OK && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = fail(MTX_ERR_HIP, "hipEventCreate failed");
  double total = 0.0;
  if (rc == MTX_OK) {
    hipGraphLaunch(exec[0], s); hipGraphLaunch(exec[1], s);          // untimed first replays
    for (int it = 0; it < iters && rc == MTX_OK; ++it)
      for (int g = 0; g < 2; ++g) {
        hipEventRecord(e0, s);
        if (hipGraphLaunch(exec[g], s) != hipSuccess) { rc = fail(MTX_ERR_HIP, "mtx_plan_time_ops: graph launch failed"); break; }
        hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(MTX_ERR_HIP, "hipEventSynchronize failed"); break; }
        float t = 0.f;
        hipEventElapsedTime(&t, e0, e1);
        total += g == 0 ? t : -t;
      }
  }
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  for (int g = 0; g < 2; ++g) { if (exec[g]) hipGraphExecDestroy(exec[g]); if (graph[g]) hipGraphDestroy(graph[g]); }
  hipStreamSynchronize(s);
  hipStreamDestroy(s);
  *ms_total = (float)total;
  return rc;
#endif
}

int mtx_plan_time(void* plan, void* stream, int iters, int use_graph, float* ms_per_iter) {
  if (!plan || !ms_per_iter || iters < 1) return fail(MTX_ERR_INVALID, "mtx_plan_time: bad arguments");
#ifdef MTX_EMU
  (void)use_graph;
  return mtx_plan_time_range(plan, 0, 1 << 30, stream, iters, ms_per_iter);
#else
  if (!use_graph) return mtx_plan_time_range(plan, 0, 1 << 30, stream, iters, ms_per_iter);
  hipStream_t s = (hipStream_t)stream;
  int rc = mtx_plan_run_graph(plan, stream);   // capture + first replay, untimed
  if (rc != MTX_OK) return rc;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(MTX_ERR_HIP, "hipEventCreate failed");
  hipEventRecord(e0, s);
  for (int i = 0; i < iters && rc == MTX_OK; ++i) rc = mtx_plan_run_graph(plan, stream);
  hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  float t = 0.f;
  if (rc == MTX_OK && e == hipSuccess) { hipEventElapsedTime(&t, e0, e1); *ms_per_iter = t / (float)iters; }
  hipEventDestroy(e0); hipEventDestroy(e1);
  if (rc != MTX_OK) return rc;
  if (e != hipSuccess) { g_err = std::string("hipEventSynchronize: ") + hipGetErrorString(e); return MTX_ERR_HIP; }
  return MTX_OK;
#endif
}

}  // extern "C"
