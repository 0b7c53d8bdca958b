// TEST INFRASTRUCTURE -- not part of the product.  Runtime of the host-side wave64 emulator (see shim/hip/hip_runtime.h).
//
// One block = blockDim.x fibers on one host thread.  Scheduling, per wave: resume every runnable lane once (a lane runs
// until it finishes or blocks in a wave-level operation or the block barrier); when no lane of the wave is runnable any more,
// the lanes blocked in a wave-level operation must all be at the same call site and are released together with the
// participant mask; when all waves of the block are finished or at the barrier, the barrier opens.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

namespace emu {
thread_local Ctx cur;

namespace {
constexpr size_t STACK_BYTES = 256 << 10;
enum State : uint8_t { READY, AT_WAVE_OP, AT_BARRIER, DONE };

// ---- fibers: a minimal x86-64 System V context switch (callee-saved registers + stack pointer) ------------------------
#if defined(__x86_64__)
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
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
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");
#else
#error "the emulator's context switch is written for x86-64 only"
#endif

std::atomic<uint64_t> g_launch_no{0};

struct Lane {
  void* sp = nullptr;
  State st = DONE;
  int site = 0;
  const void* ra = nullptr;
  uint64_t in = 0;
  Idx tid{0, 0, 0};
};

struct Block {                      // per host thread, reused for every block that thread runs
  std::vector<Lane> lanes;
  std::vector<char*> stacks;
  std::vector<uint64_t> res;        // 64 words per wave
  std::vector<uint64_t> mask;       // participant mask per wave
  std::vector<unsigned char> lds;
  void* sched_sp = nullptr;
  int running = -1;
  const std::function<void()>* fn = nullptr;
  Ctx base;
  ~Block() { for (char* s : stacks) munmap(s, STACK_BYTES); }
};
thread_local Block* tb = nullptr;

// The running lane has blocked (or finished): hand the processor straight to the next runnable lane of its wave, if there is one
// (half the context switches of going through the scheduler every time); the last one returns to the scheduler, which resolves.
void yield_to_scheduler() {
  Block* b = tb;
  const int i = b->running;
  Lane& l = b->lanes[i];
  const int hi = std::min<int>((int)b->lanes.size(), (i | 63) + 1);
  for (int j = i + 1; j < hi; j++) {
    if (b->lanes[j].st != READY) continue;
    b->running = j;
    cur = b->base; cur.tid = b->lanes[j].tid;
    emu_switch(&l.sp, b->lanes[j].sp);
    return;
  }
  emu_switch(&l.sp, b->sched_sp);
}

void fiber_main() {
  Block* b = tb;
  (*b->fn)();
  b->lanes[b->running].st = DONE;
  yield_to_scheduler();
  fprintf(stderr, "emu: resumed a finished lane\n");
  abort();
}
extern "C" void emu_fiber_entry() { fiber_main(); }

void prepare(Block* b, uint32_t i) {
  if (b->stacks.size() <= i) {
    while (b->stacks.size() <= i) {
      void* m = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (m == MAP_FAILED) { perror("emu: mmap stack"); abort(); }
      b->stacks.push_back((char*)m);
    }
  }
  // initial frame: six callee-saved registers (zero) + return address; after `ret` the stack pointer must be 8 mod 16
  uintptr_t top = ((uintptr_t)b->stacks[i] + STACK_BYTES) & ~(uintptr_t)15;
  uint64_t* sp = (uint64_t*)(top - 8);          // slot that `ret` leaves behind it: rsp = top - 8  (== 8 mod 16) on entry
  *--sp = (uint64_t)(uintptr_t)&emu_fiber_entry;
  for (int k = 0; k < 6; k++) *--sp = 0;
  b->lanes[i].sp = sp;
  b->lanes[i].st = READY;
}

void resume(Block* b, uint32_t i) {
  b->running = (int)i;
  cur = b->base;
  cur.tid = b->lanes[i].tid;
  emu_switch(&b->sched_sp, b->lanes[i].sp);
  b->running = -1;
}

[[noreturn]] void die(const Block* b, uint32_t wave, const char* what) {
  fprintf(stderr, "emu: %s (block %u, wave %u, kernel launch #%llu)\n", what, b->base.bid.x, wave, (unsigned long long)g_launch_no.load());
  const uint32_t lo = wave * 64, hi = (uint32_t)std::min<size_t>(b->lanes.size(), lo + 64);
  std::vector<int> seen;
  for (uint32_t l = lo; l < hi; l++) {
    if (b->lanes[l].st != AT_WAVE_OP) continue;
    const int sid = b->lanes[l].site;
    const void* s = b->lanes[l].ra;
    if (std::find(seen.begin(), seen.end(), sid) != seen.end()) continue;
    seen.push_back(sid);
    uint64_t m = 0;
    for (uint32_t k = lo; k < hi; k++) if (b->lanes[k].st == AT_WAVE_OP && b->lanes[k].site == sid) m |= 1ull << (k - lo);
    Dl_info di; memset(&di, 0, sizeof di);
    if (s && dladdr(s, &di) && di.dli_fbase)
      fprintf(stderr, "  lanes %016llx wait at %s+0x%zx  (addr2line -Cfie %s 0x%zx)\n", (unsigned long long)m, di.dli_sname ? di.dli_sname : "?", (size_t)((const char*)s - (const char*)di.dli_fbase), di.dli_fname, (size_t)((const char*)s - (const char*)di.dli_fbase) - 1);
    else fprintf(stderr, "  lanes %016llx wait at %p\n", (unsigned long long)m, s);
  }
  uint64_t mb = 0, md = 0;
  for (uint32_t k = lo; k < hi; k++) { if (b->lanes[k].st == AT_BARRIER) mb |= 1ull << (k - lo); if (b->lanes[k].st == DONE) md |= 1ull << (k - lo); }
  fprintf(stderr, "  lanes %016llx at the block barrier, %016llx finished\n", (unsigned long long)mb, (unsigned long long)md);
  abort();
}

std::atomic<bool> g_lax{false};

void run_block(Block* b, const std::function<void()>& fn, const Ctx& base, size_t shmem) {
  const uint32_t T = base.bdim.x * base.bdim.y * base.bdim.z;
  const uint32_t W = (T + 63) / 64;
  b->fn = &fn; b->base = base;
  b->lanes.resize(T); b->res.resize((size_t)W * 64); b->mask.resize(W);
  if (b->lds.size() < shmem + 64) b->lds.resize(shmem + 64);
  for (uint32_t i = 0; i < T; i++) {
    b->lanes[i].tid = Idx{i % base.bdim.x, (i / base.bdim.x) % base.bdim.y, i / (base.bdim.x * base.bdim.y)};
    prepare(b, i);
  }
  for (;;) {
    bool any_barrier = false, any_alive = false;
    for (uint32_t w = 0; w < W; w++) {
      const uint32_t lo = w * 64, hi = std::min(T, lo + 64);
      for (;;) {
        for (uint32_t i = lo; i < hi; i++) if (b->lanes[i].st == READY) resume(b, i);
        uint64_t at_op = 0, at_bar = 0;
        int site = 0; bool have_site = false, mixed = false;
        for (uint32_t i = lo; i < hi; i++) {
          const Lane& l = b->lanes[i];
          if (l.st == AT_WAVE_OP) {
            at_op |= 1ull << (i - lo);
            if (l.site) { if (!have_site) { site = l.site; have_site = true; } else if (site != l.site) mixed = true; }
          } else if (l.st == AT_BARRIER) at_bar |= 1ull << (i - lo);
        }
        if (!at_op) break;
        if (at_bar) die(b, w, "some lanes of a wave wait at __syncthreads while others wait in a wave-level operation");
        if (mixed && !g_lax.load(std::memory_order_relaxed)) die(b, w, "lanes of one wave wait in DIFFERENT wave-level operations (shuffle/ballot under divergent control flow)");
        for (uint32_t i = lo; i < hi; i++) if ((at_op >> (i - lo)) & 1) { b->res[i] = b->lanes[i].in; b->lanes[i].st = READY; }
        b->mask[w] = at_op;
      }
      for (uint32_t i = lo; i < hi; i++) { if (b->lanes[i].st == AT_BARRIER) any_barrier = true; if (b->lanes[i].st != DONE) any_alive = true; }
    }
    if (!any_alive) break;
    if (!any_barrier) { fprintf(stderr, "emu: scheduler stuck\n"); abort(); }
    for (uint32_t i = 0; i < T; i++) if (b->lanes[i].st == AT_BARRIER) b->lanes[i].st = READY;
  }
}

// Blocks (with their fiber stacks) are recycled across launches
std::mutex g_free_mu;
std::vector<Block*> g_free;
struct BlockLease {
  Block* b;
  BlockLease() { std::lock_guard<std::mutex> g(g_free_mu); if (g_free.empty()) b = new Block(); else { b = g_free.back(); g_free.pop_back(); } }
  ~BlockLease() { std::lock_guard<std::mutex> g(g_free_mu); g_free.push_back(b); }
};

struct Pool {
  std::vector<std::thread> th;
  unsigned n = 1;
  Pool() {
    const char* e = getenv("SMR_EMU_THREADS");
    n = e ? (unsigned)std::max(1, atoi(e)) : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char* l = getenv("SMR_EMU_LAX")) g_lax = atoi(l) != 0;
  }
};
Pool& pool() { static Pool p; return p; }
}  // namespace

uint64_t wave_exchange(uint64_t v, const uint64_t** res, int site, const void* ra) {
  Block* b = tb;
  const int i = b->running;
  Lane& l = b->lanes[i];
  l.in = v; l.site = site; l.ra = ra; l.st = AT_WAVE_OP;
  yield_to_scheduler();
  const uint32_t w = (uint32_t)i / 64;
  *res = b->res.data() + (size_t)w * 64;
  return b->mask[w];
}

void block_barrier() {
  Block* b = tb;
  b->lanes[b->running].st = AT_BARRIER;
  yield_to_scheduler();
}

void* dyn_lds() {
  uintptr_t p = (uintptr_t)tb->lds.data();
  return (void*)((p + 63) & ~(uintptr_t)63);
}

unsigned long long clock() { return (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count(); }

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& fn) {
  const uint64_t nblk = (uint64_t)grid.x * grid.y * grid.z;
  if (nblk == 0 || block.x * block.y * block.z == 0) return;
  g_launch_no++;
  std::atomic<uint64_t> next{0};
  auto worker = [&]() {
    BlockLease lease;
    Block& blk = *lease.b;
    tb = &blk;
    for (;;) {
      const uint64_t k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= nblk) break;
      Ctx base;
      base.tid = Idx{0, 0, 0};
      base.bid = Idx{(uint32_t)(k % grid.x), (uint32_t)((k / grid.x) % grid.y), (uint32_t)(k / ((uint64_t)grid.x * grid.y))};
      base.bdim = Idx{block.x, block.y, block.z};
      base.gdim = Idx{grid.x, grid.y, grid.z};
      run_block(&blk, fn, base, shmem);
    }
    tb = nullptr;
  };
  const unsigned nt = (unsigned)std::min<uint64_t>(pool().n, nblk);
  if (nt <= 1) { worker(); return; }
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nt; t++) th.emplace_back(worker);
  worker();
  for (auto& x : th) x.join();
}
}  // namespace emu

// ---- runtime API ------------------------------------------------------------------------------------------------------
struct emu_stream { int unused; };
struct emu_event { std::chrono::steady_clock::time_point t; };

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof *p);
  snprintf(p->name, sizeof p->name, "wave64 host emulator");
  snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950-emu");
  p->totalGlobalMem = (size_t)8 << 30;
  const char* e = getenv("SMR_EMU_CUS");
  p->multiProcessorCount = e ? atoi(e) : 4;
  p->warpSize = 64; p->sharedMemPerBlock = 160 << 10;
  return hipSuccess;
}
// Device allocations end right before an inaccessible guard page (up to 255 bytes of slack for the 256-byte alignment) and start
// after one: a kernel that runs past a buffer faults at once instead of corrupting a neighbour.  SMR_EMU_GUARD=0: plain malloc.
namespace {
struct GuardInfo { void* map; size_t map_n; };
std::mutex g_guard_mu;
std::vector<std::pair<void*, GuardInfo>> g_guard;
bool guard_on() { static const bool on = !(getenv("SMR_EMU_GUARD") && atoi(getenv("SMR_EMU_GUARD")) == 0); return on; }
}  // namespace
hipError_t hipMalloc(void** p, size_t n) {
  if (!guard_on()) {
    void* m = nullptr;
    if (posix_memalign(&m, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    memset(m, 0xA5, n);
    *p = m;
    return hipSuccess;
  }
  const size_t page = 4096, body = ((n ? n : 1) + 255 + page - 1) / page * page;
  char* m = (char*)mmap(nullptr, body + 2 * page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (m == MAP_FAILED) return hipErrorOutOfMemory;
  mprotect(m, page, PROT_NONE);
  mprotect(m + page + body, page, PROT_NONE);
  char* user = m + page + body - n;
  user = (char*)((uintptr_t)user & ~(uintptr_t)255);
  memset(user, 0xA5, n);           // fresh device memory is not zero: expose reads of uninitialised buffers
  { std::lock_guard<std::mutex> g(g_guard_mu); g_guard.push_back({user, GuardInfo{m, body + 2 * page}}); }
  *p = user;
  return hipSuccess;
}
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  if (!guard_on()) { free(p); return hipSuccess; }
  std::lock_guard<std::mutex> g(g_guard_mu);
  for (size_t i = 0; i < g_guard.size(); i++)
    if (g_guard[i].first == p) { munmap(g_guard[i].second.map, g_guard[i].second.map_n); g_guard[i] = g_guard.back(); g_guard.pop_back(); return hipSuccess; }
  fprintf(stderr, "emu: hipFree of an unknown pointer %p\n", p);
  abort();
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emu_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = (size_t)6 << 30; *t = (size_t)8 << 30; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipErrorInvalidValue"; }
