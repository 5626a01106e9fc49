// TEST INFRASTRUCTURE — a small SIMT emulator for the kernel sources of etl_amd/csrc.
//
// Purpose: run the *same* .hip sources (kernels.hip, fused.hip, cells.hip, scan.hip, copy.hip) and host.cpp on
// the CPU of a box without a GPU, so that the logic of a kernel change can be checked against the oracle
// before it costs GPU minutes. The sources are compiled with g++ against this header instead of
// <hip/hip_runtime.h>; every lane of a workgroup is a fiber (ucontext), workgroups run one after the
// other in blockIdx order — or, with ETLG_SIMT_GRID=<n>, n of them resident at a time and interleaved
// (simt.cpp: what the look-back kernels need to be tested at all) —, and wave / workgroup collectives (__syncthreads, __ballot, __shfl*, DPP,
// readlane, readfirstlane) are rendezvous points resolved by the scheduler in simt.cpp.
//
// What it is NOT: it is not a backend of the product. libetl_gfx950.so never contains it,
// etl_amd/native.py refuses to load a library that exports etlg_simt_marker unless the test harness
// says so, and nothing here models timing, memory ordering, LDS capacity or data races between
// waves — the GPU parity suite (-m gpu) stays the gate for every change.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <tuple>
#include <type_traits>
#include <utility>

#define ETLG_SIMT 1

// ---------------------------------------------------------------- language keywords
#define __global__ static
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local   /* one copy per resident workgroup (simt.cpp: a worker thread each with ETLG_SIMT_GRID) */
#define DEV_NOINLINE static __attribute__((noinline))
#define ETLG_DYNAMIC_LDS(name) uint8_t* const name = simt::dyn_lds()
#define ETLG_CONST_AS   /* one address space on the host */
// LDS-DMA: lane l of the wave copies 16 bytes to (base + 16 l); nothing to wait for afterwards
#define ETLG_GLDS16(g, l) memcpy((uint8_t*)(l) + 16 * (simt::g_view->tid & 63u), (const void*)(g), 16)
#define ETLG_VMEM_WAIT() ((void)0)
#define ETLG_LDS_AS
#define ETLG_LDS_LD32(p) (*(const uint32_t*)(p))
// pairs of look-back words: two plain loads / stores (one lane runs at a time)
#define ETLG_LD_PAIR(ptr, a, l) do { (a) = ((const unsigned long long*)(ptr))[0]; (l) = ((const unsigned long long*)(ptr))[1]; } while (0)
#define ETLG_ST_PAIR(ptr, a, l) do { ((unsigned long long*)(ptr))[0] = (a); ((unsigned long long*)(ptr))[1] = (l); } while (0)

struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace simt {

enum Op : int { OP_BAR = 1, OP_JOIN, OP_BALLOT, OP_ALL, OP_ANY, OP_READFIRST, OP_READLANE, OP_SHFL, OP_SHFL_UP, OP_SHFL_XOR, OP_DPP, OP_FENCE, OP_YIELD };

struct Idx { uint32_t x, y, z; };
struct LaneView { uint32_t tid, bid, bdim, gdim, gx; };   // bid: linear workgroup index over a grid of gx columns (blockIdx.x) x gdim / gx rows (blockIdx.y)
extern LaneView* g_view;   // the lane that is running
uint8_t* dyn_lds();
// rendezvous of the running lane; returns the lane's result
uint64_t collective(int op, uint64_t a, uint64_t b, uint64_t c, const void* site);
void run_grid(uint32_t grid, uint32_t block, size_t lds_bytes, void (*body)(void*), void* arg, uint32_t gx = 0);
uint64_t ticks();
void yield_hint();   // a poll loop's s_sleep: lets another resident workgroup run (ETLG_SIMT_GRID)

static inline Idx tidx() { return Idx{g_view->tid, 0, 0}; }
static inline Idx bidx() { return Idx{g_view->bid % g_view->gx, g_view->bid / g_view->gx, 0}; }
static inline Idx bdim() { return Idx{g_view->bdim, 1, 1}; }
static inline Idx gdim() { return Idx{g_view->gx, g_view->gdim / g_view->gx, 1}; }

// ---- streams (simt.cpp). Default: every enqueued operation runs at once, in program order (what the emulator always did).
// ETLG_SIMT_STREAMS=lazy: the operations of a stream — kernel launches, asynchronous copies, memsets — run as LATE as the HIP ordering rules
// allow: when the host synchronises with the stream or with an event recorded behind them, or when another stream that is being run
// reaches a wait for such an event. Work that the host code forgot to order (a kernel reading an upload on another stream without a
// wait for it, a buffer handed to a second batch while the first one's kernel is still queued, a result read before its copy was waited
// for) then really happens in the wrong order and shows up as a wrong result, instead of being hidden by immediate execution.
struct Stream;
struct Event;
bool streams_lazy();
int streams_mode();                               // 0 immediate, 1 lazy, 2 random
void stream_enqueue(void* stream, void (*fn)(void*), void* arg, void (*drop)(void*));   // fn(arg) in stream order; drop(arg) frees the closure
void stream_sync(void* stream);
void stream_destroy(void* stream);
void device_sync();                               // every stream (hipFree / hipHostFree / hipDeviceSynchronize)
void* event_create();
void event_destroy(void* ev);
void event_record(void* ev, void* stream);
void event_sync(void* ev);
void stream_wait_event(void* stream, void* ev);
void host_register(void* p, size_t n);            // pinned host memory: copies to / from it are truly asynchronous
void host_unregister(void* p);
bool host_is_pinned(const void* p);
void memcpy_async(void* d, const void* s, size_t n, int kind, void* stream);
void memset_async(void* d, int v, size_t n, void* stream);

template <class... KArgs, class... Args>
void launch(void (*k)(KArgs...), dim3 g, dim3 b, size_t lds, void* stream, Args&&... args) {
  using Tup = std::tuple<std::decay_t<KArgs>...>;
  struct Ctx { void (*k)(KArgs...); Tup t; uint32_t grid, block, gx; size_t lds; };
  Ctx* c = new Ctx{k, Tup(std::forward<Args>(args)...), g.x * g.y, b.x, g.x, lds};   // the arguments are captured by value at launch, as on the GPU
  stream_enqueue(stream, [](void* p) { Ctx* c = (Ctx*)p; run_grid(c->grid, c->block, c->lds, [](void* q) { Ctx* c2 = (Ctx*)q; std::apply(c2->k, c2->t); }, c, c->gx); },
                 c, [](void* p) { delete (Ctx*)p; });
}

}  // namespace simt

#define threadIdx (simt::tidx())
#define blockIdx (simt::bidx())
#define blockDim (simt::bdim())
#define gridDim (simt::gdim())
#define hipLaunchKernelGGL(k, g, b, lds, stream, ...) simt::launch(k, g, b, lds, (void*)(stream), __VA_ARGS__)

// ---------------------------------------------------------------- collectives
// Every collective is a function-like macro that stamps its expansion with __COUNTER__: the rendezvous is keyed
// by SOURCE site, which survives whatever the host compiler does to the call (g++ duplicates calls into both
// arms of a branch, so a return address does not identify a site).
#define SIMT_ID ((const void*)(((uintptr_t)(__COUNTER__ + 1) << 32) | (uintptr_t)__LINE__))   // order of appearance, source line
namespace simt {
static inline void c_bar(const void* id) { collective(OP_BAR, 1, 0, 0, id); }
static inline int c_bar_and(const void* id, int p) { return (int)collective(OP_BAR, p != 0, 0, 0, id); }
static inline void c_join(const void* id) { collective(OP_JOIN, 0, 0, 0, id); }
static inline void c_fence(const void* id) { collective(OP_FENCE, 0, 0, 0, id); }
static inline unsigned long long c_ballot(const void* id, int p) { return collective(OP_BALLOT, p != 0, 0, 0, id); }
static inline int c_all(const void* id, int p) { return (int)collective(OP_ALL, p != 0, 0, 0, id); }
static inline int c_any(const void* id, int p) { return (int)collective(OP_ANY, p != 0, 0, 0, id); }
static inline int c_readfirstlane(const void* id, int v) { return (int)(uint32_t)collective(OP_READFIRST, (uint32_t)v, 0, 0, id); }
static inline int c_readlane(const void* id, int v, int lane) { return (int)(uint32_t)collective(OP_READLANE, (uint32_t)v, (uint32_t)lane, 0, id); }
static inline int c_dpp(const void* id, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  return (int)(uint32_t)collective(OP_DPP, (uint32_t)src, (uint32_t)old,
                                   (uint32_t)ctrl | ((uint32_t)row_mask << 16) | ((uint32_t)bank_mask << 20) | ((uint32_t)bound_ctrl << 24), id);
}
template <class T> static inline T c_shuffle(const void* id, int op, T v, uint32_t arg) {
  static_assert(sizeof(T) <= 8, "shuffles move at most 8 bytes");
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  raw = collective(op, raw, arg, 0, id);
  T out; memcpy(&out, &raw, sizeof(T)); return out;
}
template <class T> static inline T c_shfl(const void* id, T v, int src, int = 64) { return c_shuffle(id, OP_SHFL, v, (uint32_t)src); }
template <class T> static inline T c_shfl_up(const void* id, T v, unsigned d, int = 64) { return c_shuffle(id, OP_SHFL_UP, v, d); }
template <class T> static inline T c_shfl_xor(const void* id, T v, int m, int = 64) { return c_shuffle(id, OP_SHFL_XOR, v, (uint32_t)m); }
}  // namespace simt
#define __syncthreads() simt::c_bar(SIMT_ID)
#define __syncthreads_and(p) simt::c_bar_and(SIMT_ID, p)
#define __threadfence_block() simt::c_fence(SIMT_ID)
// Reconvergence point of a divergent region that contains wave collectives. The hardware reconverges there by
// itself (the macro is empty in the HIP build); the emulator cannot see the control-flow graph, so lanes that
// skipped the region wait here for the lanes inside it instead of running ahead to the next collective.
#define ETLG_WAVE_JOIN() simt::c_join(SIMT_ID)
#define ETLG_WAVE_PRIO(n) ((void)0)   // issue priority: timing only
#define ETLG_SCALAR_COPY(dst, src) ((dst) = (src))
#define __ballot(p) simt::c_ballot(SIMT_ID, p)
#define __all(p) simt::c_all(SIMT_ID, p)
#define __any(p) simt::c_any(SIMT_ID, p)
#define __builtin_amdgcn_readfirstlane(v) simt::c_readfirstlane(SIMT_ID, v)
#define __builtin_amdgcn_readlane(v, l) simt::c_readlane(SIMT_ID, v, l)
#define __builtin_amdgcn_update_dpp(...) simt::c_dpp(SIMT_ID, __VA_ARGS__)
#define __shfl(...) simt::c_shfl(SIMT_ID, __VA_ARGS__)
#define __shfl_up(...) simt::c_shfl_up(SIMT_ID, __VA_ARGS__)
#define __shfl_xor(...) simt::c_shfl_xor(SIMT_ID, __VA_ARGS__)

// ---------------------------------------------------------------- lane-local intrinsics and atomics (one lane runs at a time)
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3u)));
}
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31u));
}
static inline void __builtin_amdgcn_s_sleep(int) { simt::yield_hint(); }
static inline unsigned long long clock64() { return simt::ticks(); }
static inline unsigned long long wall_clock64() { return simt::ticks(); }
template <class T, class U> static inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class U> static inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))

// ---------------------------------------------------------------- the slice of the HIP runtime API host.cpp uses ("device" memory = host memory)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct simt_stream* hipStream_t;
typedef struct simt_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipInit(unsigned) { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "simt error"; }
#ifndef SIMT_MALLOC_SLACK
#define SIMT_MALLOC_SLACK 256   /* the sanitizer build (tools/simt_sanitize.py) allocates exactly what was asked for */
#endif
// (ETLG_SIMT_MALLOC_POISON=1: device memory starts as 0xCD bytes instead of zeros — hipMalloc promises nothing, and a pool that hands a block
// to its second user certainly holds old data)
static inline hipError_t hipMalloc(void** p, size_t n) {
  static const bool poison = getenv("ETLG_SIMT_MALLOC_POISON") != nullptr;
  *p = poison ? malloc(n + SIMT_MALLOC_SLACK) : calloc(1, n + SIMT_MALLOC_SLACK);
  if (*p && poison) memset(*p, 0xCD, n + SIMT_MALLOC_SLACK);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { simt::device_sync(); free(p); return hipSuccess; }   // (hipFree synchronises the device)
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = calloc(1, n + (SIMT_MALLOC_SLACK ? 64 : 0)); if (*p) simt::host_register(*p, n + (SIMT_MALLOC_SLACK ? 64 : 0)); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostFree(void* p) { simt::device_sync(); simt::host_unregister(p); free(p); return hipSuccess; }
// hipMemcpy: the copy runs on the null stream and the host waits for it (the library's streams are non-blocking: nothing else is waited for)
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { simt::memcpy_async(d, s, n, (int)k, nullptr); simt::stream_sync(nullptr); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr) { simt::memcpy_async(d, s, n, (int)k, (void*)st); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr) { simt::memset_async(d, v, n, (void*)st); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { simt::stream_destroy((void*)s); free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { simt::stream_sync((void*)s); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)simt::event_create(); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr) { simt::event_record((void*)e, (void*)s); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)simt::event_create(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t e) { simt::event_sync((void*)e); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { simt::stream_wait_event((void*)s, (void*)e); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { simt::event_destroy((void*)e); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 1; return hipSuccess; }
