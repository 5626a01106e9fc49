// TEST INFRASTRUCTURE — scheduler of the SIMT emulator (see simt.h).
#include "simt.h"

#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace simt {

namespace {

enum State : uint8_t { RUN = 0, WAIT = 1, DONE = 2 };
constexpr size_t kStack = 256 * 1024;
constexpr uint32_t kMaxLanes = 1024;

struct Lane {
  LaneView view;
  ucontext_t ctx;
  State state;
  int op;
  uint64_t a, b, c, out;
  const void* site;
};

// Everything one RESIDENT workgroup owns: its lanes (fibers + their stacks), its LDS, the context of its scheduler loop. The default
// run has one (workgroups one after the other on the calling thread); ETLG_SIMT_GRID=<n> keeps n of them resident, one per worker
// thread (see Pool below) — the kernels' `__shared__` arrays are thread_local statics, so each resident workgroup has its own.
struct BlockCtx {
  Lane* lanes = nullptr;
  char* stacks = nullptr;
  uint8_t* lds = nullptr;
  ucontext_t sched;
  std::vector<uint32_t> order;   // lane order of the current scheduling round (ETLG_SIMT_ORDER)
  uint32_t bid = 0;
  void alloc() {
    if (lanes) return;
    lanes = new Lane[kMaxLanes];
    stacks = (char*)mmap(nullptr, kStack * kMaxLanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    lds = (uint8_t*)aligned_alloc(64, 160 * 1024 + 4096);
  }
};

BlockCtx g_main;            // the calling thread's workgroup (default mode)
BlockCtx* g_b = nullptr;    // the workgroup that is running (exactly one thread runs at any time)
Lane* g_cur = nullptr;
void (*g_body)(void*) = nullptr;
void* g_arg = nullptr;
uint64_t g_ticks = 0;
bool g_trace = false;
bool g_multi = false;       // a launch with several resident workgroups is running: s_sleep is a scheduling point

uint32_t g_block = 0, g_grid = 0, g_gx = 0;

// ETLG_SIMT_WATCHDOG=<seconds>: a lane that never reaches a rendezvous (a spin on a flag another lane would set)
// hangs the emulator; the alarm reports where every lane is and aborts.
void watchdog(int) {
  fprintf(stderr, "simt: watchdog — running lane %u of block %u; lanes waiting (tid: op @line):", g_cur ? g_cur->view.tid : ~0u, g_cur ? g_cur->view.bid : ~0u);
  for (uint32_t t = 0; g_b && t < g_block; t++) {
    const Lane& L = g_b->lanes[t];
    if (L.state == WAIT && (L.op != OP_BAR || t % 64 == 0)) fprintf(stderr, " %u:%d@%u(a=%llx)", t, L.op, (unsigned)((uintptr_t)L.site & 0xFFFFFFFFu), (unsigned long long)L.a);
  }
  fprintf(stderr, "\n");
  void* bt[48];
  const int n = backtrace(bt, 48);
  backtrace_symbols_fd(bt, n, 2);
  _exit(97);
}

void tramp() {
  g_body(g_arg);
  g_cur->state = DONE;
  swapcontext(&g_cur->ctx, &g_b->sched);
}

// Resolves one wave-level collective for the lanes `grp` (all of one wave, all at the same site and op).
void resolve_wave(Lane* wave0, const std::vector<int>& grp) {
  uint64_t active = 0;
  for (int l : grp) active |= 1ull << l;
  const int op = wave0[grp[0]].op;
  auto act = [&](int l) { return l >= 0 && l < 64 && ((active >> l) & 1); };
  uint64_t ballot = 0;
  for (int l : grp) if (wave0[l].a) ballot |= 1ull << l;
  for (int l : grp) {
    Lane& L = wave0[l];
    switch (op) {
      case OP_BALLOT: L.out = ballot; break;
      case OP_ALL: L.out = ballot == active; break;
      case OP_ANY: L.out = ballot != 0; break;
      case OP_FENCE: L.out = 0; break;
      case OP_READFIRST: L.out = wave0[grp[0]].a; break;  // grp is sorted by lane
      case OP_READLANE: { const int s = (int)(L.b & 63); L.out = act(s) ? wave0[s].a : 0; break; }
      case OP_SHFL: { const int s = (int)(L.b & 63); L.out = act(s) ? wave0[s].a : 0; break; }
      case OP_SHFL_UP: { const int s = l - (int)L.b; L.out = s < 0 ? L.a : (act(s) ? wave0[s].a : 0); break; }
      case OP_SHFL_XOR: { const int s = l ^ (int)(L.b & 63); L.out = act(s) ? wave0[s].a : 0; break; }
      case OP_DPP: {
        const uint32_t ctrl = (uint32_t)L.c & 0x1FFu, row_mask = ((uint32_t)L.c >> 16) & 0xFu, bank_mask = ((uint32_t)L.c >> 20) & 0xFu;
        const bool bound = ((uint32_t)L.c >> 24) & 1u;
        const int row = l >> 4, bank = (l >> 2) & 3;
        int src = -1;  // -1: no valid source lane
        bool write = ((row_mask >> row) & 1u) && ((bank_mask >> bank) & 1u);
        if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; if ((l & 15) >= n) src = l - n; }          // row_shr:n
        else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; if ((l & 15) + n <= 15) src = l + n; }  // row_shl:n
        else if (ctrl == 0x138) { if (l >= 1) src = l - 1; }                                                          // wave_shr:1
        else if (ctrl == 0x130) { if (l <= 62) src = l + 1; }                                                         // wave_shl:1
        else if (ctrl == 0x142) { if (row >= 1) src = (row - 1) * 16 + 15; }                                          // row_bcast:15
        else if (ctrl == 0x143) { if (row >= 2) src = 31; }                                                           // row_bcast:31
        else if (ctrl >= 0x150 && ctrl <= 0x15F) src = row * 16 + (int)(ctrl - 0x150);                                 // row_newbcast:n (gfx90a+)
        else if (ctrl >= 0x121 && ctrl <= 0x12F) src = row * 16 + (((l & 15) - (int)(ctrl - 0x120)) & 15);             // row_ror:n
        else { fprintf(stderr, "simt: DPP control 0x%x is not modelled\n", ctrl); abort(); }
        if (ctrl == 0x142 && row == 0) write = false;
        if (ctrl == 0x143 && row < 2) write = false;
        if (!write) L.out = L.b;
        else if (src >= 0 && act(src)) L.out = wave0[src].a;
        else L.out = bound ? 0 : L.b;
        break;
      }
      default: fprintf(stderr, "simt: unknown wave op %d\n", op); abort();
    }
  }
  for (int l : grp) wave0[l].state = RUN;
}

}  // namespace

LaneView* g_view = nullptr;
uint8_t* dyn_lds() { return g_b->lds; }
uint64_t ticks() { return ++g_ticks; }

uint64_t collective(int op, uint64_t a, uint64_t b, uint64_t c, const void* site) {
  Lane* L = g_cur;
  L->op = op; L->a = a; L->b = b; L->c = c; L->site = site; L->state = WAIT;
  swapcontext(&L->ctx, &g_b->sched);
  return L->out;
}

// s_sleep: the body of every poll loop (look-back words, late carries). With several workgroups resident it hands the processor to
// another workgroup — the one that is being waited for must get to run; on its own (default mode) there is nobody to wait for.
namespace { bool lazy_poll_wanted(); }
void yield_hint() { if (g_multi || lazy_poll_wanted()) collective(OP_YIELD, 0, 0, 0, nullptr); }

namespace {

void init_block(BlockCtx& B, uint32_t bid) {
  B.alloc();
  B.bid = bid;
  // ETLG_SIMT_LDS_POISON=1: a workgroup starts with garbage in its dynamic LDS, as on the GPU (the emulator's buffer otherwise holds what
  // the previous workgroup left — often exactly the values a kernel that forgot to initialise something hopes for)
  static const bool poison = getenv("ETLG_SIMT_LDS_POISON") != nullptr;
  if (poison) {
    static uint64_t r = 0xD1B54A32D192ED03ull;
    uint64_t* w = (uint64_t*)B.lds;
    for (size_t i = 0; i < (160 * 1024) / 8; i++) { r ^= r << 13; r ^= r >> 7; r ^= r << 17; w[i] = r; }
  }
  for (uint32_t t = 0; t < g_block; t++) {
    Lane& L = B.lanes[t];
    L.view = LaneView{t, bid, g_block, g_grid, g_gx};
    L.state = RUN;
    getcontext(&L.ctx);
    L.ctx.uc_stack.ss_sp = B.stacks + (size_t)t * kStack;
    L.ctx.uc_stack.ss_size = kStack;
    L.ctx.uc_link = nullptr;
    makecontext(&L.ctx, tramp, 0);
  }
}

// One scheduling round of the running workgroup (g_b == &B): every runnable lane to its next rendezvous (or to the end of the kernel),
// then one round of releases. Returns false once every lane is done. *yielded: a lane asked for another workgroup to run (yield_hint).
bool step_block(BlockCtx& B, bool* yielded) {
  const uint32_t block = g_block, bid = B.bid;
  const uint32_t nwaves = (block + 63) / 64;
  Lane* const lanes = B.lanes;
  // The order in which the lanes of a workgroup run between two rendezvous is not defined on the GPU; ETLG_SIMT_ORDER=reverse | shuffle
  // makes the emulator take another one than 0, 1, 2, ... (a kernel whose result depends on it has a race: the bytea[] walker of
  // round 3 had one that only the MI355X showed).
  bool ran = false;
  static const char* order_env = getenv("ETLG_SIMT_ORDER");
  static uint64_t order_rng = 0x9E3779B97F4A7C15ull;
  std::vector<uint32_t>& ord = B.order;
  if (ord.size() != block) { ord.resize(block); for (uint32_t t = 0; t < block; t++) ord[t] = t; }
  if (order_env && order_env[0] == 'r') { for (uint32_t t = 0; t < block; t++) ord[t] = block - 1 - t; }
  else if (order_env && order_env[0] == 's') {
    for (uint32_t t = block; t > 1; t--) { order_rng = order_rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(ord[t - 1], ord[(uint32_t)((order_rng >> 33) % t)]); }
  }
  for (uint32_t ti = 0; ti < block; ti++) {
    const uint32_t t = ord[ti];
    Lane& L = lanes[t];
    if (L.state != RUN) continue;
    g_cur = &L; g_view = &L.view;
    swapcontext(&B.sched, &L.ctx);
    ran = true;
  }
  // lanes inside a poll loop: they go on in the next round (no release in this one — the rest of their wave may be parked at a
  // collective these lanes have yet to reach), after the workgroups they wait for have had the processor
  bool any_yield = false;
  for (uint32_t t = 0; t < block; t++) if (lanes[t].state == WAIT && lanes[t].op == OP_YIELD) { lanes[t].state = RUN; lanes[t].out = 0; any_yield = true; }
  if (any_yield) { if (yielded) *yielded = true; return true; }
  uint32_t live = 0, at_bar = 0;
  for (uint32_t t = 0; t < block; t++) { live += lanes[t].state != DONE; at_bar += lanes[t].state == WAIT && lanes[t].op == OP_BAR; }
  if (!live) return false;
  // wave-level rendezvous: a group = the waiting lanes of one wave at one call site. When a wave has
  // several groups (divergent control flow) the one whose site comes first in the source goes first.
  bool released = false;
  std::vector<int> grp;
  for (uint32_t w = 0; w < nwaves; w++) {
    Lane* wave0 = lanes + 64 * w;
    const uint32_t n = std::min<uint32_t>(64, block - 64 * w);
    // regular collectives first (lowest source site), then — once no lane of the wave is inside a divergent
    // region any more — the lanes parked at a reconvergence point (ETLG_WAVE_JOIN)
    const void* best = nullptr;
    for (int pass = 0; pass < 2 && !best; pass++)
      for (uint32_t l = 0; l < n; l++) {
        const Lane& L = wave0[l];
        if (L.state == WAIT && L.op != OP_BAR && (L.op == OP_JOIN) == (pass == 1) && (!best || L.site < best)) best = L.site;
      }
    if (!best) continue;
    grp.clear();
    for (uint32_t l = 0; l < n; l++) if (wave0[l].state == WAIT && wave0[l].op != OP_BAR && wave0[l].site == best) grp.push_back((int)l);
    if (g_trace && wave0[grp[0]].op != OP_JOIN) {
      uint32_t waiting = 0;
      for (uint32_t l = 0; l < n; l++) waiting += wave0[l].state == WAIT && wave0[l].op != OP_BAR && wave0[l].op != OP_JOIN;
      if (waiting != grp.size()) fprintf(stderr, "simt: block %u wave %u: divergent rendezvous, %zu of %u lanes at line %u (op %d)\n", bid, w, grp.size(), waiting, (unsigned)((uintptr_t)best & 0xFFFFFFFFu), wave0[grp[0]].op);
    }
    if (wave0[grp[0]].op == OP_JOIN) { for (int l : grp) { wave0[l].out = 0; wave0[l].state = RUN; } released = true; continue; }
    resolve_wave(wave0, grp);
    released = true;
  }
  if (released) return true;
  // no wave-level rendezvous pending: everybody alive must be at the workgroup barrier
  if (at_bar == live) {
    uint64_t all = 1;
    for (uint32_t t = 0; t < block; t++) if (lanes[t].state == WAIT) all &= lanes[t].a;
    for (uint32_t t = 0; t < block; t++) if (lanes[t].state == WAIT) { lanes[t].out = all; lanes[t].state = RUN; }
    return true;
  }
  if (!ran) { fprintf(stderr, "simt: deadlock in block %u (%u live, %u at the barrier)\n", bid, live, at_bar); abort(); }
  return true;
}

// ---- several workgroups resident (ETLG_SIMT_GRID=<n>, n >= 2; ETLG_SIMT_GRID_ORDER=shuffle (default) | reverse; ETLG_SIMT_SEED)
// The GPU keeps many workgroups of a launch resident and runs them in no particular order: a tile of a look-back kernel reads words
// its predecessors have NOT written yet (what the buffer held before — another launch's words), folds aggregates instead of inclusive
// prefixes, polls. One workgroup after the other in blockIdx order shows none of that (a stale-descriptor defect of the
// boundary scan lived through two rounds of green emulator runs). Here n worker threads hold one resident workgroup each; workgroups
// become resident in blockIdx order (as the dispatcher issues them), exactly ONE thread runs at any time (the emulated memory model
// stays sequential: no fences are modelled), and the processor changes hands where a workgroup polls (s_sleep), where it ends and —
// in shuffle order — after a random number of scheduling rounds. reverse: the newest resident workgroup that is not waiting runs
// first, so every look-back finds its predecessors as late as the protocol allows. A workgroup that polled runs again only after
// another one has made progress; when all are waiting the oldest one runs (everything before it has completed).
struct Worker {
  std::thread th;
  BlockCtx bc;
  bool has_block = false, started = false, skip = false;
};
struct Pool {
  std::mutex m;
  std::condition_variable cv;
  std::vector<Worker*> w;
  int turn = -1;          // index of the worker that runs; -1: the launching thread
  uint32_t next_bid = 0, done = 0;
  bool reverse = false;
  uint64_t rng = 0x2545F4914F6CDD1Dull;
  uint64_t switches = 0, polls = 0, launches = 0;
  uint32_t draw(uint32_t n) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)((rng >> 20) % n); }
  int pick() {   // (lock held)
    int best = -1, oldest = -1;
    uint32_t cands = 0;
    for (size_t i = 0; i < w.size(); i++) {
      if (!w[i]->has_block) continue;
      if (oldest < 0 || w[i]->bc.bid < w[oldest]->bc.bid) oldest = (int)i;
      if (w[i]->skip) continue;
      cands++;
      if (reverse) { if (best < 0 || w[i]->bc.bid > w[best]->bc.bid) best = (int)i; }
      else if (draw(cands) == 0) best = (int)i;   // reservoir: uniform over the candidates
    }
    if (best >= 0) return best;
    if (oldest < 0) return -1;   // nothing left: back to the launching thread
    for (Worker* x : w) x->skip = false;
    return oldest;
  }
};
Pool* g_pool = nullptr;

void worker_main(Pool* P, Worker* self, int idx) {
  Worker& me = *self;   // (not P->w[idx]: the vector may be growing while this thread starts)
  for (;;) {
    std::unique_lock<std::mutex> lk(P->m);
    P->cv.wait(lk, [&] { return P->turn == idx; });
    lk.unlock();
    g_b = &me.bc;
    if (!me.started) { init_block(me.bc, me.bc.bid); me.started = true; }
    const uint32_t quantum = P->reverse ? 0u : 1u + P->draw(24);
    bool alive = true, yielded = false;
    for (uint32_t r = 0;; r++) {
      alive = step_block(me.bc, &yielded);
      if (!alive || yielded) break;
      if (quantum && r + 1 >= quantum) break;
    }
    lk.lock();
    if (!alive) {
      P->done++;
      for (Worker* x : P->w) x->skip = false;
      if (P->next_bid < g_grid) { me.bc.bid = P->next_bid++; me.started = false; }
      else me.has_block = false;
    } else if (yielded) { me.skip = true; P->polls++; }
    else for (Worker* x : P->w) x->skip = false;
    P->switches++;
    P->turn = P->pick();
    lk.unlock();
    P->cv.notify_all();
  }
}

void run_grid_resident(uint32_t nres) {
  if (!g_pool) {
    g_pool = new Pool();
    const char* o = getenv("ETLG_SIMT_GRID_ORDER");
    g_pool->reverse = o && o[0] == 'r';
    if (const char* sd = getenv("ETLG_SIMT_SEED")) g_pool->rng ^= (uint64_t)strtoull(sd, nullptr, 0) * 0x9E3779B97F4A7C15ull;
    if (getenv("ETLG_SIMT_GRID_STATS"))   // how much interleaving a run saw: launches with several resident workgroups, hand-overs, polls that found nothing
      atexit([] { fprintf(stderr, "simt: %llu launches with resident workgroups, %llu hand-overs, %llu after a poll\n", (unsigned long long)g_pool->launches, (unsigned long long)g_pool->switches, (unsigned long long)g_pool->polls); });
  }
  Pool* P = g_pool;
  {
    std::unique_lock<std::mutex> lk(P->m);
    while (P->w.size() < nres) {
      Worker* x = new Worker();
      P->w.push_back(x);
      x->th = std::thread(worker_main, P, x, (int)P->w.size() - 1);
      x->th.detach();
    }
    P->next_bid = 0; P->done = 0; P->launches++;
    for (size_t i = 0; i < P->w.size(); i++) {
      Worker& x = *P->w[i];
      x.skip = false; x.started = false;
      x.has_block = i < nres && P->next_bid < g_grid;
      if (x.has_block) x.bc.bid = P->next_bid++;
    }
    g_multi = true;
    P->turn = P->pick();
  }
  P->cv.notify_all();
  {
    std::unique_lock<std::mutex> lk(P->m);
    P->cv.wait(lk, [&] { return P->turn == -1; });
    g_multi = false;
    if (P->done != g_grid) { fprintf(stderr, "simt: %u of %u workgroups ran\n", P->done, g_grid); abort(); }
  }
}

}  // namespace

namespace { void lazy_poll_progress(); extern uint32_t g_depth; extern bool g_resident_mode; BlockCtx* depth_ctx(); }

void run_grid(uint32_t grid, uint32_t block, size_t lds_bytes, void (*body)(void*), void* arg, uint32_t gx) {
  if (!gx) gx = grid ? grid : 1;
  if (block > kMaxLanes) { fprintf(stderr, "simt: block of %u lanes\n", block); abort(); }
  static bool inited = false;
  static uint32_t resident = 1;
  if (!inited) {
    inited = true;
    g_trace = getenv("ETLG_SIMT_TRACE") != nullptr;
    if (const char* g = getenv("ETLG_SIMT_GRID")) resident = (uint32_t)std::max(1, atoi(g));
    g_resident_mode = resident >= 2;
    if (const char* wd = getenv("ETLG_SIMT_WATCHDOG")) { signal(SIGALRM, watchdog); alarm((unsigned)atoi(wd)); }
    if (getenv("ETLG_SIMT_SEGV")) {   // a fault inside an emulated kernel: where every lane stands + a backtrace (on a stack of its own: the lanes' stacks are small)
      static char alt[1 << 16];
      stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0; sigaltstack(&ss, nullptr);
      struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_handler = watchdog; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr);
    }
  }
  (void)lds_bytes;
  g_body = body; g_arg = arg; g_block = block; g_grid = grid; g_gx = gx;
  if (resident >= 2 && grid >= 2 && g_depth == 0) run_grid_resident(std::min(resident, grid));
  else {
    // workgroups one after the other, in blockIdx order, on the thread that runs this launch (the calling thread; a helper thread for a
    // launch that runs while another one is suspended in a poll: lazy streams, below)
    BlockCtx& B = *depth_ctx();
    g_b = &B;
    for (uint32_t bid = 0; bid < grid; bid++) {
      init_block(B, bid);
      bool polled = false;
      while (step_block(B, &polled)) {
        if (polled) { polled = false; lazy_poll_progress(); g_b = &B; }
      }
    }
  }
  g_cur = nullptr; g_view = nullptr; g_b = nullptr;
}

// ---------------------------------------------------------------- streams and events (see simt.h)
namespace {
struct StreamQ;
struct QOp { void (*fn)(void*); void* arg; void (*drop)(void*); StreamQ* wait_s; uint64_t wait_seq; uint64_t ordinal; };   // wait_s: the operation is a wait for wait_s to complete wait_seq operations
struct StreamQ {
  std::deque<QOp> q;
  uint64_t enq = 0, completed = 0;   // operations enqueued / completed so far
  bool running = false;              // an operation of this stream is executing (or suspended in a poll): the next one must not start
};
struct Ev { StreamQ* s = nullptr; uint64_t seq = 0; };
std::map<void*, StreamQ*> g_streams;   // by hipStream_t (nullptr = the null stream)
std::vector<StreamQ*> g_all;           // every queue ever made (handles of destroyed streams are forgotten, their queues stay: events point at them)
std::map<uintptr_t, size_t> g_pinned;  // hipHostMalloc blocks: start -> bytes
uint64_t g_lazy_ops = 0, g_forced_by_wait = 0, g_poll_progress = 0, g_ordinal = 0, g_running_ordinal = 0;
uint32_t g_depth = 0;                  // launches suspended in a poll below the one that is running
bool g_resident_mode = false;
size_t g_rr = 0;

bool g_random_streams = false;   // ETLG_SIMT_STREAMS=random: lazy, plus — at every enqueue — a few operations of randomly chosen streams run early
uint64_t g_srng = 0x853C49E6748FEA9Bull;
// debugging aid: a region of "device" memory whose content is compared after every operation the emulator executes (simt_watch below)
const uint8_t* g_watch = nullptr; size_t g_watch_n = 0; std::vector<uint8_t> g_watch_last;
void watch_check(const char* what, const void* fn) {
  if (!g_watch) return;
  if (memcmp(g_watch, g_watch_last.data(), g_watch_n) != 0) {
    fprintf(stderr, "simt: watched region %p changed by %s", (const void*)g_watch, what);
    if (fn) { void* a[1] = {(void*)fn}; char** sy = backtrace_symbols(a, 1); if (sy) { fprintf(stderr, " %s", sy[0]); free(sy); } }
    fprintf(stderr, " (first qword now %llu; operation number %llu)\n", (unsigned long long)*(const unsigned long long*)g_watch, (unsigned long long)g_running_ordinal);
    memcpy(g_watch_last.data(), g_watch, g_watch_n);
  }
}
bool lazy_mode() {
  static const bool on = [] {
    const char* e = getenv("ETLG_SIMT_STREAMS");
    const bool l = e && (e[0] == 'l' || e[0] == 'r');
    g_random_streams = e && e[0] == 'r';
    if (const char* sd = getenv("ETLG_SIMT_SEED")) g_srng ^= (uint64_t)strtoull(sd, nullptr, 0) * 0x9E3779B97F4A7C15ull;
    if (l && getenv("ETLG_SIMT_GRID_STATS"))
      atexit([] { fprintf(stderr, "simt: %llu operations ran deferred, %llu forced by another stream's wait, %llu while a kernel of another stream polled\n",
                          (unsigned long long)g_lazy_ops, (unsigned long long)g_forced_by_wait, (unsigned long long)g_poll_progress); });
    return l;
  }();
  return on;
}
StreamQ* sq(void* stream) {
  auto it = g_streams.find(stream);
  if (it != g_streams.end()) return it->second;
  StreamQ* S = new StreamQ();
  g_all.push_back(S);
  return g_streams[stream] = S;
}

// A launch that runs while others are suspended needs lanes, stacks, LDS and `__shared__` arrays of its own: depth d > 0 runs on helper
// thread d (thread_local statics) with block context d. One thread runs at any time.
struct Helper {
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  void (*fn)(void*) = nullptr; void* arg = nullptr;
  bool busy = false;
  BlockCtx bc;
};
std::vector<Helper*> g_helpers;   // [depth]; [0] only lends its block context to the calling thread
BlockCtx* depth_ctx() {
  if (g_depth == 0) return &g_main;
  return &g_helpers[g_depth]->bc;
}
void helper_main(Helper* h) {
  for (;;) {
    std::unique_lock<std::mutex> lk(h->m);
    h->cv.wait(lk, [&] { return h->busy; });
    lk.unlock();
    h->fn(h->arg);
    lk.lock();
    h->busy = false;
    lk.unlock();
    h->cv.notify_all();
  }
}
void exec_on_depth(void (*fn)(void*), void* arg) {
  if (g_depth == 0) { fn(arg); return; }
  while (g_helpers.size() <= g_depth) {
    Helper* h = new Helper();
    g_helpers.push_back(h);
    if (g_helpers.size() > 1) { h->th = std::thread(helper_main, h); h->th.detach(); }
  }
  Helper* h = g_helpers[g_depth];
  { std::unique_lock<std::mutex> lk(h->m); h->fn = fn; h->arg = arg; h->busy = true; }
  h->cv.notify_all();
  std::unique_lock<std::mutex> lk(h->m);
  h->cv.wait(lk, [&] { return !h->busy; });
}

bool force(StreamQ* S, uint64_t n);
// Runs the oldest operation of T if it can run now. false: T is empty, busy, or its oldest operation waits for a stream that is busy.
bool try_run_front(StreamQ* T) {
  if (T->running || T->q.empty()) return false;
  QOp op = T->q.front();
  if (op.wait_s) {
    if (op.wait_s->completed < op.wait_seq) {
      if (op.wait_s->running) return false;
      g_forced_by_wait++;
      T->running = true;
      const bool ok = force(op.wait_s, op.wait_seq);
      T->running = false;
      if (!ok) return false;
    }
    T->q.pop_front();
    T->completed++;
    return true;
  }
  T->q.pop_front();
  T->running = true;
  g_lazy_ops++;
  g_running_ordinal = op.ordinal;
  exec_on_depth(op.fn, op.arg);
  watch_check("a queued operation", (const void*)op.fn);
  if (op.drop) op.drop(op.arg);
  T->running = false;
  T->completed++;
  return true;
}
// Runs S until it has completed n operations. false: it cannot get there now (S or something it waits for is busy further up the stack).
bool force(StreamQ* S, uint64_t n) {
  while (S->completed < n) if (!try_run_front(S)) return false;
  return true;
}
void force_or_die(StreamQ* S, uint64_t n, const char* what) {
  if (force(S, n)) return;
  fprintf(stderr, "simt: %s cannot complete: the stream (or one it waits for) is busy — a cycle of waits, or a synchronisation from inside a kernel\n", what);
  abort();
}

bool lazy_poll_wanted() {
  if (!lazy_mode() || g_resident_mode) return false;
  for (StreamQ* T : g_all) if (!T->running && !T->q.empty()) return true;
  return false;
}
// A kernel polls for something another stream's work produces (the late carry of a batch that runs BESIDE its predecessor, plan.hip /
// lookback.hip.h): whatever is enqueued on the other streams may legally run now. One operation of one of them, round robin.
void lazy_poll_progress() {
  if (!lazy_mode() || g_resident_mode || g_all.empty()) return;
  void (*sv_body)(void*) = g_body; void* sv_arg = g_arg;
  const uint32_t sv_block = g_block, sv_grid = g_grid, sv_gx = g_gx;
  BlockCtx* sv_b = g_b; Lane* sv_cur = g_cur; LaneView* sv_view = g_view;
  g_depth++;
  for (size_t k = 0; k < g_all.size(); k++) {
    StreamQ* T = g_all[(g_rr + k) % g_all.size()];
    if (try_run_front(T)) { g_rr = (g_rr + k + 1) % g_all.size(); g_poll_progress++; break; }
  }
  g_depth--;
  g_body = sv_body; g_arg = sv_arg; g_block = sv_block; g_grid = sv_grid; g_gx = sv_gx; g_b = sv_b; g_cur = sv_cur; g_view = sv_view;
}
}  // namespace

bool streams_lazy() { return lazy_mode(); }
int streams_mode() { return !lazy_mode() ? 0 : g_random_streams ? 2 : 1; }

void stream_enqueue(void* stream, void (*fn)(void*), void* arg, void (*drop)(void*)) {
  if (!lazy_mode()) { fn(arg); watch_check("an operation", (const void*)fn); if (drop) drop(arg); return; }
  StreamQ* S = sq(stream);
  S->q.push_back(QOp{fn, arg, drop, nullptr, 0, ++g_ordinal});
  S->enq++;
  if (g_random_streams && g_depth == 0) {   // any schedule between "at once" and "as late as possible": some of what is queued runs now
    auto draw = [] { g_srng ^= g_srng << 13; g_srng ^= g_srng >> 7; g_srng ^= g_srng << 17; return (uint32_t)(g_srng >> 24); };
    for (uint32_t k = draw() % 4; k > 0 && !g_all.empty(); k--) (void)try_run_front(g_all[draw() % g_all.size()]);
  }
}
void stream_sync(void* stream) { if (lazy_mode()) { StreamQ* S = sq(stream); force_or_die(S, S->enq, "hipStreamSynchronize"); } }
void stream_destroy(void* stream) {
  if (!lazy_mode()) return;
  auto it = g_streams.find(stream);
  if (it == g_streams.end()) return;
  force_or_die(it->second, it->second->enq, "hipStreamDestroy");   // (hipStreamDestroy lets the queued work finish)
  g_streams.erase(it);
}
void device_sync() {
  if (!lazy_mode()) return;
  for (size_t i = 0; i < g_all.size(); i++) force_or_die(g_all[i], g_all[i]->enq, "a device synchronisation (hipFree / hipHostFree)");
}
void* event_create() { return new Ev(); }
void event_destroy(void* ev) { delete (Ev*)ev; }
void event_record(void* ev, void* stream) {
  if (!ev) return;
  Ev* e = (Ev*)ev;
  if (!lazy_mode()) { e->s = nullptr; return; }
  e->s = sq(stream); e->seq = e->s->enq;   // complete once everything enqueued on the stream so far has run
}
void event_sync(void* ev) { if (lazy_mode() && ev && ((Ev*)ev)->s) force_or_die(((Ev*)ev)->s, ((Ev*)ev)->seq, "hipEventSynchronize"); }
void stream_wait_event(void* stream, void* ev) {
  if (!lazy_mode() || !ev) return;
  Ev* e = (Ev*)ev;
  if (!e->s) return;   // never recorded: no dependency (HIP's rule)
  StreamQ* S = sq(stream);
  if (S == e->s) return;   // (its own earlier work: stream order already says so)
  // the wait refers to the record that was LAST made when the wait is enqueued, not to later ones
  S->q.push_back(QOp{nullptr, nullptr, nullptr, e->s, e->seq, ++g_ordinal});
  S->enq++;
}
void host_register(void* p, size_t n) { g_pinned[(uintptr_t)p] = n; }
void host_unregister(void* p) { g_pinned.erase((uintptr_t)p); }
bool host_is_pinned(const void* p) {
  auto it = g_pinned.upper_bound((uintptr_t)p);
  if (it == g_pinned.begin()) return false;
  --it;
  return (uintptr_t)p < it->first + it->second;
}
// Copies. A copy between device memory and PINNED host memory is asynchronous in earnest: it reads its source when it runs. From
// PAGEABLE host memory the runtime stages the bytes before the call returns (the caller may reuse the buffer at once), and a copy INTO
// pageable memory returns only when it is complete — the emulator cannot tell device from pageable host memory by address, so the
// direction comes from `kind` (1 = host to device, 2 = device to host).
void memcpy_async(void* d, const void* s, size_t n, int kind, void* stream) {
  if (!lazy_mode() || !n) { if (n) memmove(d, s, n); return; }
  struct C { void* d; const void* s; size_t n; void* staged; };
  C* c = new C{d, s, n, nullptr};
  if (kind == 1 && !host_is_pinned(s)) { c->staged = malloc(n); memcpy(c->staged, s, n); c->s = c->staged; }
  stream_enqueue(stream, [](void* p) { C* c = (C*)p; memmove(c->d, c->s, c->n); }, c, [](void* p) { C* c = (C*)p; free(c->staged); delete c; });
  if (kind == 2 && !host_is_pinned(d)) stream_sync(stream);
}
void memset_async(void* d, int v, size_t n, void* stream) {
  if (!lazy_mode()) { memset(d, v, n); return; }
  struct M { void* d; int v; size_t n; };
  stream_enqueue(stream, [](void* p) { M* m = (M*)p; memset(m->d, m->v, m->n); }, new M{d, v, n}, [](void* p) { delete (M*)p; });
}

}  // namespace simt

extern "C" int etlg_simt_marker(void) { return 1; }
// debugging aid for a host-side probe (never called by the product sources): report every executed operation that changes [p, p + n)
extern "C" unsigned long long simt_op_ordinal(void) { return simt::g_ordinal; }   // number of the operation enqueued last (lazy streams)
extern "C" void simt_watch(const void* p, size_t n) {
  simt::g_watch = (const uint8_t*)p; simt::g_watch_n = p ? n : 0;
  simt::g_watch_last.assign((const uint8_t*)p, (const uint8_t*)p + simt::g_watch_n);
}
