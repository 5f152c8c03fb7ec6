// TEST INFRASTRUCTURE ONLY -- scheduler of the host-side SIMT emulator (see hip/hip_runtime.h).
// One ucontext fiber per GPU thread, blocks executed one after another, cooperative scheduling:
// every runnable fiber runs until it reaches a block barrier, a wave-collective, or exits; a
// barrier is released when all live fibers of its scope wait on it; no progress = deadlock = abort.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

namespace emul {
enum State { RUNNABLE = 0, WAIT_BLOCK = 1, WAIT_WAVE = 2, DONE = 3 };
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  Idx tid{0, 0, 0};
  int lin = 0;
  int state = DONE;
};

static constexpr size_t STACK = 256 * 1024;
static std::vector<Fiber*> pool;
static ucontext_t main_ctx;
static Fiber* current = nullptr;
static const std::function<void()>* body_fn = nullptr;
static Idx g_bid, g_bdim, g_gdim;
static int g_nthreads = 0;
static std::vector<char> g_dyn;
static std::vector<uint64_t> g_slots;   // [wave][lane][2]

Fiber* cur() { return current; }
const Idx& tid() { return current->tid; }
const Idx& bid() { return g_bid; }
const Idx& bdim() { return g_bdim; }
const Idx& gdim() { return g_gdim; }
int lane() { return current->lin & 63; }
char* dyn_smem() { return g_dyn.data(); }
uint64_t* wave_slot(int l, int which) { return &g_slots[(size_t)((current->lin >> 6) * 64 + l) * 2 + which]; }
bool lane_alive(int l) {
  int lin = (current->lin & ~63) + l;
  return lin < g_nthreads && pool[lin]->state != DONE;
}

static void yield_to_main(State s) {
  current->state = s;
  Fiber* me = current;
  swapcontext(&me->ctx, &main_ctx);
}
void sync_block() { yield_to_main(WAIT_BLOCK); }
void sync_wave() { yield_to_main(WAIT_WAVE); }

static void trampoline() {
  (*body_fn)();
  current->state = DONE;
  Fiber* me = current;
  swapcontext(&me->ctx, &main_ctx);
}

static void run_block(unsigned nthreads) {
  for (unsigned t = 0; t < nthreads; ++t) {
    Fiber* f = pool[t];
    getcontext(&f->ctx);
    f->ctx.uc_stack.ss_sp = f->stack;
    f->ctx.uc_stack.ss_size = STACK;
    f->ctx.uc_link = &main_ctx;
    makecontext(&f->ctx, trampoline, 0);
    f->lin = (int)t;
    f->tid = {t % g_bdim.x, (t / g_bdim.x) % g_bdim.y, t / (g_bdim.x * g_bdim.y)};
    f->state = RUNNABLE;
  }
  const unsigned nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool ran = false;
    for (unsigned t = 0; t < nthreads; ++t) {
      Fiber* f = pool[t];
      if (f->state != RUNNABLE) continue;
      current = f;
      swapcontext(&main_ctx, &f->ctx);
      ran = true;
    }
    // release wave barriers
    bool released = false;
    unsigned live = 0, at_block = 0;
    for (unsigned w = 0; w < nwaves; ++w) {
      unsigned lo = w * 64, hi = lo + 64 < nthreads ? lo + 64 : nthreads;
      unsigned wl = 0, ww = 0;
      for (unsigned t = lo; t < hi; ++t) {
        if (pool[t]->state != DONE) ++wl;
        if (pool[t]->state == WAIT_WAVE) ++ww;
        if (pool[t]->state == WAIT_BLOCK) ++at_block;
      }
      live += wl;
      if (ww && ww == wl) {
        for (unsigned t = lo; t < hi; ++t)
          if (pool[t]->state == WAIT_WAVE) pool[t]->state = RUNNABLE;
        released = true;
      }
    }
    if (live == 0) return;
    if (!released && at_block == live) {
      for (unsigned t = 0; t < nthreads; ++t)
        if (pool[t]->state == WAIT_BLOCK) pool[t]->state = RUNNABLE;
      released = true;
    }
    if (!released && !ran) {
      fprintf(stderr, "emul: DEADLOCK in block (%u,%u,%u): divergent barrier / wave collective\n", g_bid.x, g_bid.y, g_bid.z);
      abort();
    }
    if (!released) {
      // somebody ran but nothing could be released: only legal if a fiber is still RUNNABLE
      bool any = false;
      for (unsigned t = 0; t < nthreads; ++t) any |= pool[t]->state == RUNNABLE;
      if (!any) {
        fprintf(stderr, "emul: DEADLOCK (mixed barrier scopes) in block (%u,%u,%u)\n", g_bid.x, g_bid.y, g_bid.z);
        abort();
      }
    }
  }
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t shmem) {
  unsigned nthreads = block.x * block.y * block.z;
  if (nthreads == 0 || nthreads > 1024) { fprintf(stderr, "emul: bad block size %u\n", nthreads); abort(); }
  while (pool.size() < nthreads) {
    Fiber* f = new Fiber();
    f->stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    pool.push_back(f);
  }
  g_nthreads = (int)nthreads;
  g_bdim = {block.x, block.y, block.z};
  g_gdim = {grid.x, grid.y, grid.z};
  g_dyn.assign(shmem + 64, (char)0xA5);   // poison: reading uninitialised LDS shows up as garbage
  g_slots.assign((size_t)((nthreads + 63) / 64) * 64 * 2, 0);
  body_fn = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_bid = {bx, by, bz};
        run_block(nthreads);
      }
  body_fn = nullptr;
}
}  // namespace emul
