// emu_runtime.cc -- fibre scheduler of the kernel emulator (TEST INFRASTRUCTURE, see shim/hip/hip_runtime.h).
//
// One workgroup at a time.  Every work-item runs on its own stack; control moves between the scheduler and a
// work-item with a hand-written x86-64 context switch (callee-saved registers + stack pointer; ucontext's
// swapcontext makes a system call per switch).  A work-item runs until it reaches a rendezvous, deposits its
// operand and yields; when the last unfinished lane of its wave arrives the rendezvous completes and every lane
// reads the table of deposits on its next turn.  Deposit tables alternate between two buffers: a lane cannot be
// two rendezvous ahead of another lane of its wave, so a table stays intact while any lane still reads it.
#include <hip/hip_runtime.h>

#include <vector>

alignas(64) unsigned char smem[160 * 1024];

// switch_to(&save_sp, load_sp): saves the callee-saved registers and the stack pointer of the caller in
// *save_sp and resumes the context whose stack pointer is load_sp.
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


namespace emu {
namespace {

constexpr size_t kStackBytes = 512 * 1024;
constexpr int kMaxThreads = 1024;

struct fibre {
  void* sp = nullptr;            // saved stack pointer while switched out
  unsigned char* stack = nullptr;
  thread_state st;
  bool done = true;
  int waiting_gen = -1;          // generation of the rendezvous this lane waits for (-1: runnable)
};

struct wave_state {
  uint64_t table[2][64];
  int line[64];
  int arrived = 0;
  int generation = 0;
  unsigned long long arrived_mask = 0, active[2] = {0, 0};
};

fibre g_fibres[kMaxThreads];
wave_state g_waves[kMaxThreads / 64];
void* g_scheduler_sp = nullptr;
fibre* g_current = nullptr;
const std::function<void()>* g_body = nullptr;
int g_threads = 0;

void yield_to_scheduler() { emu_switch(&g_current->sp, g_scheduler_sp); }

void fibre_main() {
  (*g_body)();
  g_current->done = true;
  // a finished lane no longer takes part: a rendezvous the others wait in may now be complete
  wave_state& w = g_waves[g_current->st.tid.x / 64];
  (void)w;
  yield_to_scheduler();
  fprintf(stderr, "emu: finished fibre resumed\n");
  abort();
}

void prepare(fibre& f) {
  if (!f.stack) f.stack = static_cast<unsigned char*>(aligned_alloc(64, kStackBytes));
  // initial frame: six callee-saved registers (popped by emu_switch), then the return address fibre_main;
  // at fibre_main's entry the stack pointer must be 8 modulo 16 (as after a call)
  uintptr_t top = reinterpret_cast<uintptr_t>(f.stack) + kStackBytes;
  top &= ~static_cast<uintptr_t>(15);
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                   // keeps the alignment: entry rsp = top - 8
  *--sp = reinterpret_cast<void*>(&fibre_main);      // return address of emu_switch's `ret`
  for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp, rbx, r12..r15
  f.sp = sp;
  f.done = false;
  f.waiting_gen = -1;
}

int unfinished_lanes(int wave, unsigned long long* mask) {
  int n = 0;
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) {
    const int t = wave * 64 + l;
    if (t < g_threads && !g_fibres[t].done) { ++n; m |= 1ull << l; }
  }
  *mask = m;
  return n;
}

}  // namespace

thread_state& self() { return g_current->st; }

const uint64_t* exchange(uint64_t in, int line, unsigned long long* active) {
  fibre* me = g_current;
  const int wave = me->st.tid.x / 64, lane = me->st.lane;
  wave_state& w = g_waves[wave];
  const int gen = w.generation, buf = gen & 1;
  w.table[buf][lane] = in;
  w.line[lane] = line;
  w.arrived += 1;
  w.arrived_mask |= 1ull << lane;
  me->waiting_gen = gen;
  while (w.generation == gen) {
    unsigned long long alive;
    const int need = unfinished_lanes(wave, &alive);
    if (w.arrived == need && w.arrived_mask == alive) {        // the last lane to arrive completes the rendezvous
      for (int l = 0; l < 64; ++l)
        if (((alive >> l) & 1ull) && w.line[l] != line) {
          fprintf(stderr, "emu: cross-lane operation under divergent control flow: lane %d is at source line %d, lane %d at line %d "
                          "(workgroup %u)\n", lane, line, l, w.line[l], me->st.bid.x);
          abort();
        }
      w.active[buf] = alive;
      w.arrived = 0;
      w.arrived_mask = 0;
      w.generation = gen + 1;
      break;
    }
    yield_to_scheduler();
  }
  me->waiting_gen = -1;
  *active = w.active[buf];
  return w.table[buf];
}

// s_barrier: a work-item waits until every unfinished work-item of the workgroup has arrived (from the same source
// line: a barrier under divergent control flow is refused like any other rendezvous).
namespace {
int g_bar_generation = 0, g_bar_arrived = 0, g_bar_line = -1;
}
void block_barrier(int line) {
  const int gen = g_bar_generation;
  if (g_bar_arrived == 0) g_bar_line = line;
  else if (g_bar_line != line) {
    fprintf(stderr, "emu: __syncthreads() under divergent control flow: source lines %d and %d\n", g_bar_line, line);
    abort();
  }
  g_bar_arrived += 1;
  while (g_bar_generation == gen) {
    int alive = 0;
    for (int t = 0; t < g_threads; ++t) alive += g_fibres[t].done ? 0 : 1;
    if (g_bar_arrived >= alive) { g_bar_arrived = 0; g_bar_generation = gen + 1; break; }
    g_current->waiting_gen = -2;              // parked in the block barrier: polled by the scheduler
    yield_to_scheduler();
  }
  g_current->waiting_gen = -1;
}

void run_grid(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes) {
  const int threads = (int)(block.x * block.y * block.z);
  if (threads > kMaxThreads || lds_bytes > sizeof(smem)) { fprintf(stderr, "emu: launch too large\n"); abort(); }
  if (g_current) { fprintf(stderr, "emu: nested launch\n"); abort(); }
  g_body = &body;
  g_threads = threads;
  for (unsigned bz = 0; bz < grid.z; ++bz)
  for (unsigned by = 0; by < grid.y; ++by)
  for (unsigned b = 0; b < grid.x; ++b) {
    g_bar_generation = 0; g_bar_arrived = 0;
    // LDS is not initialised on the device: a garbage pattern (SWB_EMU_LDS_FILL overrides the byte, to check that no
    // result depends on what a previous workgroup left there)
    static const int fill = getenv("SWB_EMU_LDS_FILL") ? (int)strtol(getenv("SWB_EMU_LDS_FILL"), nullptr, 0) : 0xA5;
    memset(smem, fill, lds_bytes ? lds_bytes : 64);
    for (int t = 0; t < threads; ++t) {
      fibre& f = g_fibres[t];
      prepare(f);
      f.st.tid = dim3((unsigned)t, 0, 0);
      f.st.bid = dim3(b, by, bz);
      f.st.bdim = block;
      f.st.gdim = grid;
      f.st.lane = t & 63;
    }
    for (int w = 0; w < (threads + 63) / 64; ++w) { g_waves[w] = wave_state(); }
    for (;;) {
      bool any = false, progressed = false;
      for (int t0 = 0; t0 < threads; ++t0) {
        // SWB_EMU_LANE_ORDER=reverse: lanes take their turns in descending order -- results must not depend on the order
        // in which lanes run between two rendezvous (a cross-lane LDS hazard without a fence would show)
        static const bool reverse = getenv("SWB_EMU_LANE_ORDER") && !strcmp(getenv("SWB_EMU_LANE_ORDER"), "reverse");
        const int t = reverse ? threads - 1 - t0 : t0;
        fibre& f = g_fibres[t];
        if (f.done) continue;
        any = true;
        if (f.waiting_gen >= 0) {
          // a lane parked in a rendezvous is resumed when that rendezvous has completed, or when lanes of its
          // wave have finished meanwhile (it re-evaluates the completion test itself)
          wave_state& w = g_waves[t / 64];
          unsigned long long alive;
          const int need = unfinished_lanes(t / 64, &alive);
          if (w.generation == f.waiting_gen && !(w.arrived == need && w.arrived_mask == alive)) continue;
        }
        g_current = &f;
        emu_switch(&g_scheduler_sp, f.sp);
        g_current = nullptr;
        progressed = true;
      }
      if (!any) break;
      if (!progressed) {
        fprintf(stderr, "emu: deadlock in workgroup %u: some lanes wait in a cross-lane operation the others never reach\n", b);
        for (int t = 0; t < threads; ++t)
          if (!g_fibres[t].done) fprintf(stderr, "  lane %d waits at source line %d\n", t, g_waves[t / 64].line[t & 63]);
        abort();
      }
    }
  }
  g_body = nullptr;
}

}  // namespace emu

// Self-test for the tests: two halves of a wave reach two DIFFERENT ballots -- the rendezvous must refuse it.
extern "C" void emu_selftest_divergent_ballot() {
  emu::run_grid([]() {
    unsigned long long m;
    if (emu::self().lane < 32)
      m = __ballot(true);
    else
      m = __ballot(false);
    (void)m;
  }, dim3(1), dim3(64), 64);
}

// Event counters of an instrumented build (SWB_EMU_STATS=1, tests/emu/build_emu.py): counted by lane 0 of a wave.
// Bounds checks the emulated build adds to the kernel source (build_emu.py: reads of the hand-off lists): violations are
// counted, not fatal, so that a test can assert on them.
static long g_emu_violations;
extern "C" void emu_check(int ok) { if (!ok) ++g_emu_violations; }
extern "C" long emu_violations(int reset) {
  const long v = g_emu_violations;
  if (reset) g_emu_violations = 0;
  return v;
}
static long g_emu_counters[32];
extern "C" void emu_count(int counter, long amount) {
  if (emu::self().lane == 0 && counter >= 0 && counter < 32) g_emu_counters[counter] += amount;
}
extern "C" long emu_stats(int counter, int reset) {
  const long v = (counter >= 0 && counter < 32) ? g_emu_counters[counter] : 0;
  if (reset) memset(g_emu_counters, 0, sizeof(g_emu_counters));
  return v;
}
