// Experiment: does a kernel launched with hipExtAnyOrderLaunch (AQL packet without the barrier bit) start while the
// previous kernel of the SAME stream is still running on gfx950?  The header documents the flag as unsupported on gfx9.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_build/exp_anyorder tools/exp_anyorder.hip && tools/_build/exp_anyorder
// Each kernel spins for a fixed time on the constant 100 MHz clock and records its first start / last end.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned long long* rec) {
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { atomicMin(&rec[0], t0); atomicMax(&rec[1], wall_clock64()); }
}

static void report(const char* what, unsigned long long* rec, int n) {
  printf("%-44s", what);
  for (int k = 0; k < n; ++k) printf("  k%d [%7.1f, %7.1f] us", k, (rec[2 * k] - rec[0]) / 100.0, (rec[2 * k + 1] - rec[0]) / 100.0);
  printf("\n");
}

int main() {
  unsigned long long *rec, *drec;        // device-side records (atomics on host memory take microseconds each), copied back
  rec = (unsigned long long*)malloc(64 * sizeof(unsigned long long));
  CHECK(hipMalloc(&drec, 64 * sizeof(unsigned long long)));
  hipStream_t s, s2;
  CHECK(hipStreamCreate(&s));
  CHECK(hipStreamCreate(&s2));
  hipEvent_t ev;
  CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const dim3 g(64), b(64);
  const long long T = 10000;   // 100 us
  auto reset = [&]() {
    for (int k = 0; k < 8; ++k) { rec[2 * k] = ~0ull; rec[2 * k + 1] = 0ull; }
    CHECK(hipMemcpy(drec, rec, 64 * sizeof(unsigned long long), hipMemcpyHostToDevice));
  };
  auto fetch = [&]() { CHECK(hipMemcpy(rec, drec, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost)); };
  for (int rep = 0; rep < 3; ++rep) {
    reset();
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec);
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec + 2);
    CHECK(hipStreamSynchronize(s));
    fetch();
    report("in order (plain launches)", rec, 2);

    reset();
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec);
    hipExtLaunchKernelGGL(spin, g, b, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, T, drec + 2);
    CHECK(hipStreamSynchronize(s));
    fetch();
    report("second with hipExtAnyOrderLaunch", rec, 2);

    reset();
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec);
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec + 2);
    hipExtLaunchKernelGGL(spin, g, b, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, T, drec + 4);
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec + 6);
    CHECK(hipStreamSynchronize(s));
    fetch();
    report("A, B, C(any order), D", rec, 4);

    reset();
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec);
    hipLaunchKernelGGL(spin, g, b, 0, s2, T, drec + 2);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipStreamSynchronize(s2));
    fetch();
    report("two streams", rec, 2);

    // fork / join through events: A on s; B on s2 after A (event); C on s after B (event)
    reset();
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec);
    CHECK(hipEventRecord(ev, s));
    CHECK(hipStreamWaitEvent(s2, ev, 0));
    hipLaunchKernelGGL(spin, g, b, 0, s2, T, drec + 2);
    CHECK(hipEventRecord(ev, s2));
    CHECK(hipStreamWaitEvent(s, ev, 0));
    hipLaunchKernelGGL(spin, g, b, 0, s, T, drec + 4);
    CHECK(hipStreamSynchronize(s));
    fetch();
    report("A(s) -event-> B(s2) -event-> C(s)", rec, 3);
  }
  // gap between back-to-back tiny kernels of one stream, with and without the barrier bit
  for (int mode = 0; mode < 2; ++mode) {
    reset();
    for (int k = 0; k < 8; ++k) {
      if (mode == 0 || k == 0) hipLaunchKernelGGL(spin, dim3(1), b, 0, s, 100, drec + 2 * k);
      else hipExtLaunchKernelGGL(spin, dim3(1), b, 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 100, drec + 2 * k);
    }
    CHECK(hipStreamSynchronize(s));
    fetch();
    report(mode ? "8 x 1 us kernels, any order" : "8 x 1 us kernels, in order", rec, 8);
  }
  return 0;
}
