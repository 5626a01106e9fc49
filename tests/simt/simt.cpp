// TEST INFRASTRUCTURE — scheduler of the SIMT emulator (see simt.h).
#include "simt.h"

#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

namespace simt {
static std::vector<uint32_t> g_order;   // lane order of the current scheduling round (ETLG_SIMT_ORDER)

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

Lane g_lanes[kMaxLanes];
ucontext_t g_sched;
Lane* g_cur = nullptr;
char* g_stacks = nullptr;
uint8_t* g_lds = nullptr;
void (*g_body)(void*) = nullptr;
void* g_arg = nullptr;
uint64_t g_ticks = 0;
bool g_trace = false;

uint32_t g_block = 0;

// ETLG_SIMT_WATCHDOG=<seconds>: a lane that never reaches a rendezvous (a spin on a flag another lane would set)
// hangs the emulator; the alarm reports where every lane is and aborts.
void watchdog(int) {
  fprintf(stderr, "simt: watchdog — running lane %u of block %u; lanes waiting (tid: op @line):", g_cur ? g_cur->view.tid : ~0u, g_cur ? g_cur->view.bid : ~0u);
  for (uint32_t t = 0; t < g_block; t++) {
    const Lane& L = g_lanes[t];
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
  swapcontext(&g_cur->ctx, &g_sched);
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
uint8_t* dyn_lds() { return g_lds; }
uint64_t ticks() { return ++g_ticks; }

uint64_t collective(int op, uint64_t a, uint64_t b, uint64_t c, const void* site) {
  Lane* L = g_cur;
  L->op = op; L->a = a; L->b = b; L->c = c; L->site = site; L->state = WAIT;
  swapcontext(&L->ctx, &g_sched);
  return L->out;
}

void run_grid(uint32_t grid, uint32_t block, size_t lds_bytes, void (*body)(void*), void* arg, uint32_t gx) {
  if (!gx) gx = grid ? grid : 1;
  if (block > kMaxLanes) { fprintf(stderr, "simt: block of %u lanes\n", block); abort(); }
  if (!g_stacks) {
    g_stacks = (char*)mmap(nullptr, kStack * kMaxLanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    g_lds = (uint8_t*)aligned_alloc(64, 160 * 1024 + 4096);
    g_trace = getenv("ETLG_SIMT_TRACE") != nullptr;
    if (const char* wd = getenv("ETLG_SIMT_WATCHDOG")) { signal(SIGALRM, watchdog); alarm((unsigned)atoi(wd)); }
    if (getenv("ETLG_SIMT_SEGV")) {   // a fault inside an emulated kernel: where every lane stands + a backtrace (on a stack of its own: the lanes' stacks are small)
      static char alt[1 << 16];
      stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof(alt); ss.ss_flags = 0; sigaltstack(&ss, nullptr);
      struct sigaction sa; memset(&sa, 0, sizeof(sa)); sa.sa_handler = watchdog; sa.sa_flags = SA_ONSTACK; sigaction(SIGSEGV, &sa, nullptr);
    }
  }
  (void)lds_bytes;
  g_body = body; g_arg = arg; g_block = block;
  const uint32_t nwaves = (block + 63) / 64;
  std::vector<int> grp;
  for (uint32_t bid = 0; bid < grid; bid++) {
    for (uint32_t t = 0; t < block; t++) {
      Lane& L = g_lanes[t];
      L.view = LaneView{t, bid, block, grid, gx};
      L.state = RUN;
      getcontext(&L.ctx);
      L.ctx.uc_stack.ss_sp = g_stacks + (size_t)t * kStack;
      L.ctx.uc_stack.ss_size = kStack;
      L.ctx.uc_link = nullptr;
      makecontext(&L.ctx, tramp, 0);
    }
    for (;;) {
      // run every runnable lane to its next rendezvous (or to the end of the kernel). The order in which the lanes of a workgroup
      // run between two rendezvous is not defined on the GPU; ETLG_SIMT_ORDER=reverse | shuffle makes the emulator take another
      // one than 0, 1, 2, ... (a kernel whose result depends on it has a race: the bytea[] walker of round 3 had one that only
      // the MI355X showed). Workgroups stay in blockIdx order: the look-back kernels wait for their predecessors.
      bool ran = false;
      static const char* order_env = getenv("ETLG_SIMT_ORDER");
      static uint64_t order_rng = 0x9E3779B97F4A7C15ull;
      std::vector<uint32_t>& ord = g_order;
      if (ord.size() != block) { ord.resize(block); for (uint32_t t = 0; t < block; t++) ord[t] = t; }
      if (order_env && order_env[0] == 'r') { for (uint32_t t = 0; t < block; t++) ord[t] = block - 1 - t; }
      else if (order_env && order_env[0] == 's') {
        for (uint32_t t = block; t > 1; t--) { order_rng = order_rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(ord[t - 1], ord[(uint32_t)((order_rng >> 33) % t)]); }
      }
      for (uint32_t ti = 0; ti < block; ti++) {
        const uint32_t t = ord[ti];
        Lane& L = g_lanes[t];
        if (L.state != RUN) continue;
        g_cur = &L; g_view = &L.view;
        swapcontext(&g_sched, &L.ctx);
        ran = true;
      }
      uint32_t live = 0, at_bar = 0;
      for (uint32_t t = 0; t < block; t++) { live += g_lanes[t].state != DONE; at_bar += g_lanes[t].state == WAIT && g_lanes[t].op == OP_BAR; }
      if (!live) break;
      // wave-level rendezvous: a group = the waiting lanes of one wave at one call site. When a wave has
      // several groups (divergent control flow) the one whose site comes first in the source goes first.
      bool released = false;
      for (uint32_t w = 0; w < nwaves; w++) {
        Lane* wave0 = g_lanes + 64 * w;
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
      if (released) continue;
      // no wave-level rendezvous pending: everybody alive must be at the workgroup barrier
      if (at_bar == live) {
        uint64_t all = 1;
        for (uint32_t t = 0; t < block; t++) if (g_lanes[t].state == WAIT) all &= g_lanes[t].a;
        for (uint32_t t = 0; t < block; t++) if (g_lanes[t].state == WAIT) { g_lanes[t].out = all; g_lanes[t].state = RUN; }
        continue;
      }
      if (!ran) { fprintf(stderr, "simt: deadlock in block %u (%u live, %u at the barrier)\n", bid, live, at_bar); abort(); }
    }
  }
  g_cur = nullptr; g_view = nullptr;
}

}  // namespace simt

extern "C" int etlg_simt_marker(void) { return 1; }
