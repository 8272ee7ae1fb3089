// emu_hip.cpp — runtime of the test-only SIMT simulator (see emu_hip.h).
#include "emu_hip.h"
#include <memory>
#include <mutex>

#if defined(__x86_64__)
// void emu_switch(Ctx* from, Ctx* to): push the callee-saved registers and the two floating-point control words, park the stack pointer in
// *from, take *to's and unwind the same frame there.  A fresh fiber's stack is laid out so that the final `ret` lands in fiber_entry.
__asm__(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch,.-emu_switch
)");
#endif

namespace emu {

thread_local Block* B = nullptr;

void fiber_entry() {
  Block* b = B;
  (*b->body)();
  b->fibers[b->cur].done = true;
  ctx_switch(b->fibers[b->cur].ctx, b->main_ctx);      // back to the scheduler for good: a finished fiber is never resumed
  abort();
}

static void prepare_fiber(Block& blk, Fiber& f) {
  f.done = false;
#if defined(__x86_64__)
  uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15);
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                       // the return address fiber_entry would return to (it never does): keeps rsp + 8 on a 16-byte boundary at entry
  *--sp = reinterpret_cast<void*>(&fiber_entry);          // popped by emu_switch's `ret`
  for (int i = 0; i < 6; ++i) *--sp = nullptr;            // rbp, rbx, r12..r15
  --sp;
  uint32_t* cw = reinterpret_cast<uint32_t*>(sp);
  cw[0] = 0x1F80u;                                        // MXCSR: default rounding, exceptions masked
  cw[1] = 0x037Fu;                                        // x87 control word (low 16 bits are loaded)
  f.ctx.sp = sp;
#else
  getcontext(&f.ctx);
  f.ctx.uc_stack.ss_sp = f.stack;
  f.ctx.uc_stack.ss_size = kStack;
  f.ctx.uc_link = &blk.main_ctx;
  makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
}

void run_block(Block& blk) {
  B = &blk;
  const int n = blk.nthreads;
  for (int i = 0; i < n; ++i) prepare_fiber(blk, blk.fibers[i]);
  blk.bar_count = 0;
  for (auto& w : blk.waves) { w.count = 0; }
  for (;;) {
    bool any = false;
    for (int i = 0; i < n; ++i) {
      if (blk.fibers[i].done) continue;
      any = true;
      blk.cur = i;
      ctx_switch(blk.main_ctx, blk.fibers[i].ctx);
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

// Fiber stacks and per-block bookkeeping are kept between launches (a launch used to allocate and zero 96 KiB per HIP thread per worker):
// a worker borrows an arena for the duration of a launch; stack pages are touched once and stay mapped.
struct Arena {
  char* stacks = nullptr;
  size_t stack_bytes = 0;
  Block blk;
  std::vector<char> smem;
  ~Arena() { free(stacks); }
};
static std::mutex g_arena_mutex;
static std::vector<std::unique_ptr<Arena>> g_free_arenas;

static std::unique_ptr<Arena> borrow_arena() {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  if (g_free_arenas.empty()) return std::unique_ptr<Arena>(new Arena());
  std::unique_ptr<Arena> a = std::move(g_free_arenas.back());
  g_free_arenas.pop_back();
  return a;
}
static void return_arena(std::unique_ptr<Arena> a) {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  g_free_arenas.push_back(std::move(a));
}

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads > kMaxThreads) { fprintf(stderr, "emu: block too large\n"); abort(); }
  const long total = (long)grid.x * grid.y * grid.z;
  std::atomic<long> next{0};
  auto worker = [&]() {
    std::unique_ptr<Arena> arena = borrow_arena();
    Block& blk = arena->blk;
    blk.bdim = block;
    blk.gdim = grid;
    blk.nthreads = nthreads;
    blk.fibers.resize(nthreads);
    blk.waves.resize((nthreads + kWave - 1) / kWave);
    blk.body = &body;
    if (arena->smem.size() < dyn_smem + 64) arena->smem.resize(dyn_smem + 64);
    blk.dyn_smem = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(arena->smem.data()) + 15) & ~uintptr_t(15));
    const size_t need = (size_t)nthreads * kStack;
    if (arena->stack_bytes < need) {
      free(arena->stacks);
      arena->stacks = static_cast<char*>(malloc(need));
      if (!arena->stacks) { fprintf(stderr, "emu: out of memory for fiber stacks\n"); abort(); }
      arena->stack_bytes = need;
    }
    for (int i = 0; i < nthreads; ++i) {
      Fiber& f = blk.fibers[i];
      f.lin = i;
      f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
      f.stack = arena->stacks + (size_t)i * kStack;
    }
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= total) break;
      blk.bid = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(blk);
    }
    return_arena(std::move(arena));
  };
  int nw = n_workers();
  if (total < nw) nw = (int)total;
  if (nw <= 1) { worker(); return; }
  std::vector<std::thread> th;
  for (int i = 0; i < nw; ++i) th.emplace_back(worker);
  for (auto& t : th) t.join();
}

}  // namespace emu
