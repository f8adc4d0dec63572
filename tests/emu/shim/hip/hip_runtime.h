// hip_runtime.h -- stand-in for the HIP runtime and device builtins, used ONLY by the kernel emulator
// (tests/emu, see tests/emu/README.md).  TEST INFRASTRUCTURE: it lets the CPU test suite execute the source of
// spriteworld_amd/csrc (the fused step kernel included) lane by lane on the host and compare it with the
// oracle where no GPU exists.  It is never part of the product: spriteworld_amd/ loads csrc/libswb.so built by
// hipcc for gfx950 and raises when that is missing.
//
// Execution model: one workgroup at a time; every work-item is a cooperative fibre (own stack, hand-written
// context switch).  A cross-lane operation (ballot, readlane, shuffle, DPP move, wave barrier) is a rendezvous
// of all unfinished lanes of the 64-lane wave: the emulator checks that all of them arrive from the SAME call
// site (the kernels keep cross-lane operations under wave-uniform control flow; anything else aborts with the
// two source lines).  LDS is one host buffer filled with a garbage pattern before each workgroup.
#ifndef SWB_EMU_HIP_RUNTIME_H_
#define SWB_EMU_HIP_RUNTIME_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

// ---------------------------------------------------------------- language keywords
#define __global__
#define __device__ inline
#define __host__
#define __shared__
#define __constant__ static
#define __forceinline__ __attribute__((always_inline))
#define __launch_bounds__(...)
#define __HIP_MEMORY_SCOPE_AGENT 4

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// ---------------------------------------------------------------- the emulator core (tests/emu/emu_runtime.cc)
namespace emu {
struct thread_state { dim3 tid, bid, bdim, gdim; int lane; };
thread_state& self();                       // the running work-item
// Rendezvous of the wave: deposits `in` (8 bytes), waits for every unfinished lane, returns the table of all
// lanes' deposits (valid until this lane's next rendezvous) and, in *active, the mask of lanes that took part.
const uint64_t* exchange(uint64_t in, int line, unsigned long long* active);
}  // namespace emu

extern unsigned char smem[];                // the workgroup's LDS
#define threadIdx (emu::self().tid)
#define blockIdx (emu::self().bid)
#define blockDim (emu::self().bdim)
#define gridDim (emu::self().gdim)

// ---------------------------------------------------------------- cross-lane operations
namespace emu {
template <typename T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
inline unsigned long long ballot(bool p, int line) {
  unsigned long long act;
  const uint64_t* t = exchange(p ? 1u : 0u, line, &act);
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) if (((act >> i) & 1ull) && t[i]) m |= 1ull << i;
  return m;
}
inline int readlane(int v, int src, int line) {        // a disabled source lane reads as 0
  unsigned long long act;
  const uint64_t* t = exchange(to_bits(v), line, &act);
  src &= 63;
  return ((act >> src) & 1ull) ? from_bits<int>(t[src]) : 0;
}
inline int readfirstlane(int v, int line) {
  unsigned long long act;
  const uint64_t* t = exchange(to_bits(v), line, &act);
  return from_bits<int>(t[__builtin_ctzll(act)]);
}
template <typename T> inline T shfl(T v, int src, int line) {
  unsigned long long act;
  const uint64_t* t = exchange(to_bits(v), line, &act);
  src &= 63;
  return ((act >> src) & 1ull) ? from_bits<T>(t[src]) : T(0);
}
template <typename T> inline T shfl_up(T v, unsigned delta, int line) {
  unsigned long long act;
  const uint64_t* t = exchange(to_bits(v), line, &act);
  const int src = self().lane - (int)delta;
  return (src >= 0 && ((act >> src) & 1ull)) ? from_bits<T>(t[src]) : v;
}
// v_mov_b32_dpp with row_mask = bank_mask = 0xf, bound_ctrl = 0: a lane without a valid source keeps `old`
inline int update_dpp(int old, int src_val, int ctrl, int line) {
  unsigned long long act;
  const uint64_t* t = exchange(to_bits(src_val), line, &act);
  const int l = self().lane;
  int s = -1;
  if (ctrl >= 0x000 && ctrl <= 0x0ff) s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);          // quad_perm
  else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; s = ((l & 15) >= n) ? l - n : -1; }   // row_shr:n
  else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl & 15; s = ((l & 15) + n < 16) ? l + n : -1; }   // row_shl:n
  else if (ctrl == 0x138) s = l - 1;                                                         // wave_shr:1
  else if (ctrl == 0x130) s = (l < 63) ? l + 1 : -1;                                         // wave_shl:1
  else { fprintf(stderr, "emu: DPP control 0x%x not implemented (line %d)\n", ctrl, line); abort(); }
  return (s >= 0 && ((act >> s) & 1ull)) ? from_bits<int>(t[s]) : old;
}
inline void wave_barrier(int line) { unsigned long long act; (void)exchange(0, line, &act); }
void block_barrier(int line);               // s_barrier: every unfinished work-item of the workgroup
}  // namespace emu

#define __ballot(p) emu::ballot((p), __LINE__)
#define __builtin_amdgcn_readlane(v, l) emu::readlane((v), (l), __LINE__)
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane((v), __LINE__)
#define __shfl(v, src, width) emu::shfl((v), (src), __LINE__)
#define __shfl_xor(v, mask, width) emu::shfl((v), emu::self().lane ^ (mask), __LINE__)
#define __shfl_up(v, delta, width) emu::shfl_up((v), (delta), __LINE__)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) emu::update_dpp((old), (src), (ctrl), __LINE__)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier(__LINE__)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __syncthreads() emu::block_barrier(__LINE__)
/* a clock that advances with every reading (the cover kernel orders the next launch by the cycles its waves took) */
#define __builtin_amdgcn_s_setprio(level) ((void)0)
inline unsigned long long __builtin_amdgcn_s_memtime() { static unsigned long long t = 0; return t += 997; }
// v_perm_b32 D = bytes of {S0, S1} selected by S2: selector 0-3 = S1's bytes, 4-7 = S0's bytes (only these are used)
inline unsigned emu_perm(unsigned s0, unsigned s1, unsigned sel) {
  const uint64_t both = ((uint64_t)s0 << 32) | s1;
  unsigned out = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned k = (sel >> (8 * i)) & 0xff;
    if (k > 7) { fprintf(stderr, "emu: v_perm_b32 selector %u not implemented\n", k); abort(); }
    out |= (unsigned)((both >> (8 * k)) & 0xff) << (8 * i);
  }
  return out;
}
#define __builtin_amdgcn_perm(s0, s1, sel) emu_perm((s0), (s1), (sel))

// ---------------------------------------------------------------- arithmetic intrinsics (this file is compiled
// with -ffp-contract=off: a + b, a * b round once, std::fma is the only fused operation)
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __ddiv_rn(double a, double b) { return a / b; }
inline double __dsqrt_rn(double a) { return std::sqrt(a); }
inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline int __mul24(int a, int b) { return (int)((uint32_t)((a << 8) >> 8) * (uint32_t)((b << 8) >> 8)); }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
using std::max;
using std::min;
inline int max(int a, unsigned b) { return a > (int)b ? a : (int)b; }
inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }

// atomics: work-items are cooperative fibres of one host thread, so plain read-modify-write is atomic
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicXor(T* p, T v) { T o = *p; *p = o ^ v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
#define __hip_atomic_load(p, order, scope) (*(p))

// ---------------------------------------------------------------- the runtime API the host side uses
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; };
inline const char* hipGetErrorString(hipError_t) { return "emulated HIP runtime"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : 2; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; ++r) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
  return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "gfx950:emulated"); p->multiProcessorCount = 256; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)64 << 30; *total_b = (size_t)288 << 30; return hipSuccess; }
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 16; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
#define hipEventDisableTiming 2
#define hipStreamNonBlocking 1
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int token; *s = &token; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

namespace emu {
void run_grid(const std::function<void()>& body, dim3 grid, dim3 block, size_t lds_bytes);
}
#define hipLaunchKernelGGL(fn, grid, block, lds, stream, ...) \
  emu::run_grid([=]() { fn(__VA_ARGS__); }, (grid), (block), (lds))

#endif  // SWB_EMU_HIP_RUNTIME_H_
