// emu_hip.cpp — runtime of the test-only SIMT simulator (see emu_hip.h).
#include "emu_hip.h"
#include <mutex>

namespace emu {

thread_local Block* B = nullptr;

void fiber_entry() {
  Block* b = B;
  (*b->body)();
  b->fibers[b->cur].done = true;
  // return to the scheduler; uc_link handles it too, but be explicit
  swapcontext(&b->fibers[b->cur].ctx, &b->main_ctx);
}

void run_block(Block& blk) {
  B = &blk;
  const int n = blk.nthreads;
  for (int i = 0; i < n; ++i) {
    Fiber& f = blk.fibers[i];
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &blk.main_ctx;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
  }
  blk.bar_count = 0;
  for (auto& w : blk.waves) { w.count = 0; }
  for (;;) {
    bool any = false;
    for (int i = 0; i < n; ++i) {
      if (blk.fibers[i].done) continue;
      any = true;
      blk.cur = i;
      swapcontext(&blk.main_ctx, &blk.fibers[i].ctx);
    }
    if (!any) break;
  }
  B = nullptr;
}

static int n_workers() {
  const char* e = getenv("MTX_EMU_THREADS");
  int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (n > 16) n = 16;
  return n;
}

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
  const long total = (long)grid.x * grid.y * grid.z;
  std::atomic<long> next{0};
  auto worker = [&]() {
    Block blk;
    blk.bdim = block;
    blk.gdim = grid;
    blk.nthreads = nthreads;
    blk.fibers.resize(nthreads);
    blk.waves.resize((nthreads + kWave - 1) / kWave);
    blk.body = &body;
    std::vector<char> smem(dyn_smem + 64);
    blk.dyn_smem = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(smem.data()) + 15) & ~uintptr_t(15));
    std::vector<char> stacks((size_t)nthreads * kStack);
    for (int i = 0; i < nthreads; ++i) {
      Fiber& f = blk.fibers[i];
      f.lin = i;
      f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
      f.stack = stacks.data() + (size_t)i * kStack;
    }
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= total) break;
      blk.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(blk);
    }
  };
  int nw = n_workers();
  if (total < nw) nw = (int)total;
  if (nw <= 1) { worker(); return; }
  std::vector<std::thread> th;
  for (int i = 0; i < nw; ++i) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

}  // namespace emu
