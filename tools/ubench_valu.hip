// ubench_valu.hip -- issue cost of the instructions the step kernel is made of (gfx950).
//
// Each test runs a loop of 16 INDEPENDENT instances (16 distinct destination registers, sources
// that no instance writes) of one instruction on every SIMD of the chip with WPS waves per SIMD,
// long enough (tens of ms) for the clock to settle.  Reported: shader cycles per wave-instruction
// per SIMD = elapsed s_memtime ticks of one wave / (iterations * 16 * WPS), the effective shader
// clock (s_memtime ticks / s_memrealtime 100 MHz ticks), and the same figure from wall time.
// Build: hipcc --offload-arch=gfx950 -O2 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) (void)(x)

// destination i = %0..%15; sources: %16,%17 ints, %18 float, %19,%20 doubles, %21 sgpr
#define OPS                                                                                      \
  "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
  "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
#define OPSD                                                                                     \
  "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), \
  "+v"(d[8]), "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15])

#define I16(PRE, POST)                                                                              \
  PRE "%0" POST "\n\t" PRE "%1" POST "\n\t" PRE "%2" POST "\n\t" PRE "%3" POST "\n\t" PRE "%4" POST "\n\t" \
  PRE "%5" POST "\n\t" PRE "%6" POST "\n\t" PRE "%7" POST "\n\t" PRE "%8" POST "\n\t" PRE "%9" POST "\n\t" \
  PRE "%10" POST "\n\t" PRE "%11" POST "\n\t" PRE "%12" POST "\n\t" PRE "%13" POST "\n\t" PRE "%14" POST "\n\t" \
  PRE "%15" POST "\n\t"

#define DEFKERNEL(NAME, PRE, POST, DOUBLE)                                                   \
  __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* out, int iters) {      \
    int r[16];                                                                               \
    double d[16];                                                                            \
    for (int i = 0; i < 16; ++i) { r[i] = threadIdx.x + i; d[i] = threadIdx.x * 0.5 + i; }   \
    int a = threadIdx.x * 3 + 1, b = blockIdx.x + 7;                                         \
    float f = threadIdx.x * 0.5f;                                                            \
    double x = 1.25 + threadIdx.x, y = 3.0;                                                  \
    int s = 3;                                                                               \
    asm volatile("" : "+v"(a), "+v"(b), "+v"(f), "+v"(x), "+v"(y), "+s"(s));                 \
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();               \
    for (int i = 0; i < iters; ++i) {                                                        \
      if (DOUBLE) asm volatile(I16(PRE, POST) : OPSD : "v"(a), "v"(b), "v"(f), "v"(x), "v"(y), "s"(s) : "vcc", "s30", "s31"); \
      else asm volatile(I16(PRE, POST) : OPS : "v"(a), "v"(b), "v"(f), "v"(x), "v"(y), "s"(s) : "vcc", "s30", "s31"); \
    }                                                                                        \
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();               \
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = w1 - w0; }         \
    int acc = 0;                                                                             \
    for (int i = 0; i < 16; ++i) acc += r[i] + (int)d[i];                                    \
    if (acc == 0x12345678) out[1] = 1;                                                       \
  }

DEFKERNEL(fma_f32, "v_fma_f32 ", ", %18, %18, %18", 0)
DEFKERNEL(mad_i32_i24, "v_mad_i32_i24 ", ", %16, %17, %16", 0)
DEFKERNEL(mad_i32_i24_sgpr, "v_mad_i32_i24 ", ", %16, %21, %17", 0)
DEFKERNEL(mul_i32_i24, "v_mul_i32_i24 ", ", %16, %17", 0)
DEFKERNEL(mul_lo_u32, "v_mul_lo_u32 ", ", %16, %17", 0)
DEFKERNEL(add_u32, "v_add_u32 ", ", %16, %17", 0)
DEFKERNEL(sub_u32, "v_sub_u32 ", ", %16, %17", 0)
DEFKERNEL(add3_u32, "v_add3_u32 ", ", %16, %17, %16", 0)
DEFKERNEL(lshl_add_u32, "v_lshl_add_u32 ", ", %16, 2, %17", 0)
DEFKERNEL(lshlrev_b32, "v_lshlrev_b32 ", ", 3, %16", 0)
DEFKERNEL(ashrrev_i32, "v_ashrrev_i32 ", ", 22, %16", 0)
DEFKERNEL(and_b32, "v_and_b32 ", ", %16, %17", 0)
DEFKERNEL(xor_b32, "v_xor_b32 ", ", %16, %17", 0)
DEFKERNEL(mov_b32, "v_mov_b32 ", ", %16", 0)
DEFKERNEL(med3_i32, "v_med3_i32 ", ", %16, %17, %16", 0)
DEFKERNEL(min_i32, "v_min_i32 ", ", %16, %17", 0)
DEFKERNEL(max_i32, "v_max_i32 ", ", %16, %17", 0)
DEFKERNEL(bfe_i32, "v_bfe_i32 ", ", %16, 10, 10", 0)
DEFKERNEL(and_or_b32, "v_and_or_b32 ", ", %16, %17, %16", 0)
DEFKERNEL(perm_b32, "v_perm_b32 ", ", %16, %17, %16", 0)
DEFKERNEL(bcnt, "v_bcnt_u32_b32 ", ", %16, %17", 0)
DEFKERNEL(ffbl, "v_ffbl_b32 ", ", %16", 0)
DEFKERNEL(cndmask, "v_cndmask_b32 ", ", %16, %17, vcc", 0)
DEFKERNEL(cmp_lt_i32, "v_cmp_lt_i32 vcc, %16, ", "", 0)
DEFKERNEL(cmp_sgpr, "v_cmp_lt_i32 s[30:31], %16, ", "", 0)
DEFKERNEL(mov_dpp, "v_mov_b32_dpp ", ", %16 row_shr:1 row_mask:0xf bank_mask:0xf", 0)
DEFKERNEL(mul_sdwa, "v_mul_i32_i24_sdwa ", ", %16, %17 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1", 0)
DEFKERNEL(dot4_i32_i8, "v_dot4_i32_i8 ", ", %16, %17, %16", 0)
DEFKERNEL(dot2_i32_i16, "v_dot2_i32_i16 ", ", %16, %17, %16", 0)
DEFKERNEL(cvt_i32_f32, "v_cvt_i32_f32 ", ", %18", 0)
DEFKERNEL(cvt_f32_i32, "v_cvt_f32_i32 ", ", %16", 0)
DEFKERNEL(floor_f32, "v_floor_f32 ", ", %18", 0)
DEFKERNEL(add_f32, "v_add_f32 ", ", %18, %18", 0)
DEFKERNEL(mul_f32, "v_mul_f32 ", ", %18, %18", 0)
DEFKERNEL(sqrt_f32, "v_sqrt_f32 ", ", %18", 0)
DEFKERNEL(rcp_f32, "v_rcp_f32 ", ", %18", 0)
DEFKERNEL(pk_mad_i16, "v_pk_mad_i16 ", ", %16, %17, %16", 0)
DEFKERNEL(pk_add_u16, "v_pk_add_u16 ", ", %16, %17", 0)
DEFKERNEL(readlane, "v_readlane_b32 s30, ", ", 3", 0)
DEFKERNEL(bpermute, "ds_bpermute_b32 ", ", %16, %17", 0)
DEFKERNEL(ds_read_b32, "ds_read_b32 ", ", %16", 0)
DEFKERNEL(add_f64, "v_add_f64 ", ", %19, %20", 1)
DEFKERNEL(mul_f64, "v_mul_f64 ", ", %19, %20", 1)
DEFKERNEL(fma_f64, "v_fma_f64 ", ", %19, %20, %19", 1)
DEFKERNEL(rcp_f64, "v_rcp_f64 ", ", %19", 1)
DEFKERNEL(sqrt_f64, "v_sqrt_f64 ", ", %19", 1)
DEFKERNEL(cvt_f64_i32, "v_cvt_f64_i32 ", ", %16", 1)
DEFKERNEL(mad_u64_u32, "v_mad_u64_u32 ", ", vcc, %16, %17, %19", 1)
// ---- round-2 additions: encodings (VOP2 4-byte vs VOP3 / SDWA / DPP 8-byte), accumulate-in-place forms
// (candidates for the vertical pass, 18 v_mad_i32_i24 per canvas row) and mixed streams
DEFKERNEL(add_u32_e64, "v_add_u32_e64 ", ", %16, %17", 0)
DEFKERNEL(add_u32_lit, "v_add_u32 ", ", 0x12345, %16", 0)
DEFKERNEL(mul_u32_u24, "v_mul_u32_u24 ", ", %16, %17", 0)
DEFKERNEL(mad_u32_u24, "v_mad_u32_u24 ", ", %16, %17, %16", 0)
DEFKERNEL(fmac_f32, "v_fmac_f32 ", ", %18, %18", 0)
DEFKERNEL(fmac_f32_sgpr, "v_fmac_f32 ", ", %21, %18", 0)
DEFKERNEL(dot2c_i32_i16, "v_dot2c_i32_i16 ", ", %16, %17", 0)
DEFKERNEL(dot2c_i32_i16_sgpr, "v_dot2c_i32_i16 ", ", %21, %17", 0)
DEFKERNEL(dot4c_i32_i8, "v_dot4c_i32_i8 ", ", %16, %17", 0)
DEFKERNEL(dot8c_i32_i4, "v_dot8c_i32_i4 ", ", %16, %17", 0)
DEFKERNEL(dot2c_f32_f16, "v_dot2c_f32_f16 ", ", %16, %17", 0)
DEFKERNEL(pk_fma_f32, "v_pk_fma_f32 ", ", %19, %20, %19", 1)
DEFKERNEL(pk_mul_lo_u16, "v_pk_mul_lo_u16 ", ", %16, %17", 0)
DEFKERNEL(mad_i32_i16, "v_mad_i32_i16 ", ", %16, %17, %16", 0)
DEFKERNEL(max_i32_e32, "v_max_i32 ", ", 0, %16", 0)
DEFKERNEL(min_u32, "v_min_u32 ", ", %16, %17", 0)
DEFKERNEL(lshlrev_b32_v, "v_lshlrev_b32 ", ", %16, %17", 0)
DEFKERNEL(lshrrev_b32, "v_lshrrev_b32 ", ", 22, %16", 0)
DEFKERNEL(ashr_sdwa, "v_ashrrev_i32_sdwa ", ", %16, %17 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD", 0)
DEFKERNEL(alignbit, "v_alignbit_b32 ", ", %16, %17, 22", 0)
DEFKERNEL(cvt_pk_u8_f32, "v_cvt_pk_u8_f32 ", ", %18, 1, %16", 0)
DEFKERNEL(sat_pk_u8_i16, "v_sat_pk_u8_i16 ", ", %16", 0)
DEFKERNEL(or_b32, "v_or_b32 ", ", %16, %17", 0)
DEFKERNEL(lshl_or_b32, "v_lshl_or_b32 ", ", %16, 16, %17", 0)
DEFKERNEL(sub_u32_sgpr, "v_sub_u32 ", ", %21, %16", 0)
DEFKERNEL(ds_read_b64, "ds_read_b64 ", ", %16", 1)
// mixed streams (16 instructions per iteration: 8 + 8 interleaved)
#define MIX8(A, B)                                                                                          \
  A "%0" B "\n\t v_add_u32 %1, %16, %17\n\t" A "%2" B "\n\t v_add_u32 %3, %16, %17\n\t" A "%4" B "\n\t v_add_u32 %5, %16, %17\n\t" \
  A "%6" B "\n\t v_add_u32 %7, %16, %17\n\t" A "%8" B "\n\t v_add_u32 %9, %16, %17\n\t" A "%10" B "\n\t v_add_u32 %11, %16, %17\n\t" \
  A "%12" B "\n\t v_add_u32 %13, %16, %17\n\t" A "%14" B "\n\t v_add_u32 %15, %16, %17\n\t"
#define DEFMIX(NAME, A, B)                                                                   \
  __global__ void __launch_bounds__(256) k_##NAME(unsigned long long* out, int iters) {      \
    int r[16];                                                                               \
    for (int i = 0; i < 16; ++i) r[i] = threadIdx.x + i;                                     \
    int a = threadIdx.x * 3 + 1, b = blockIdx.x + 7;                                         \
    float f = threadIdx.x * 0.5f;                                                            \
    double x = 1.25 + threadIdx.x, y = 3.0;                                                  \
    int s = 3;                                                                               \
    asm volatile("" : "+v"(a), "+v"(b), "+v"(f), "+v"(x), "+v"(y), "+s"(s));                 \
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();               \
    for (int i = 0; i < iters; ++i)                                                          \
      asm volatile(MIX8(A, B) : OPS : "v"(a), "v"(b), "v"(f), "v"(x), "v"(y), "s"(s) : "vcc", "s30", "s31"); \
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();               \
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = w1 - w0; }         \
    int acc = 0;                                                                             \
    for (int i = 0; i < 16; ++i) acc += r[i];                                                \
    if (acc == 0x12345678) out[1] = 1;                                                       \
  }
DEFMIX(mix_mad_add, "v_mad_i32_i24 ", ", %16, %21, %17")
DEFMIX(mix_salu_add, "s_add_u32 s30, s30, 1 ;", "")
DEFMIX(mix_med3_add, "v_med3_i32 ", ", %16, %17, %16")
DEFMIX(mix_dsread_add, "ds_read_b32 ", ", %16")

struct test { const char* name; void (*fn)(unsigned long long*, int); };
#define T(NAME) {#NAME, k_##NAME}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200000;
  const test tests[] = {T(fma_f32), T(mad_i32_i24), T(mad_i32_i24_sgpr), T(mul_i32_i24), T(mul_lo_u32), T(add_u32), T(sub_u32),
                        T(add3_u32), T(lshl_add_u32), T(lshlrev_b32), T(ashrrev_i32), T(and_b32), T(xor_b32), T(mov_b32),
                        T(med3_i32), T(min_i32), T(max_i32), T(bfe_i32), T(and_or_b32), T(perm_b32), T(bcnt), T(ffbl),
                        T(cndmask), T(cmp_lt_i32), T(cmp_sgpr), T(mov_dpp), T(mul_sdwa), T(dot4_i32_i8), T(dot2_i32_i16),
                        T(cvt_i32_f32), T(cvt_f32_i32), T(floor_f32), T(add_f32), T(mul_f32), T(sqrt_f32), T(rcp_f32),
                        T(pk_mad_i16), T(pk_add_u16), T(readlane), T(bpermute), T(ds_read_b32), T(add_f64), T(mul_f64),
                        T(fma_f64), T(rcp_f64), T(sqrt_f64), T(cvt_f64_i32), T(mad_u64_u32),
                        // round-2 additions (run alone with: ubench_valu ITERS new)
                        T(add_u32_e64), T(add_u32_lit), T(mul_u32_u24), T(mad_u32_u24), T(fmac_f32), T(fmac_f32_sgpr),
                        T(dot2c_i32_i16), T(dot2c_i32_i16_sgpr), T(dot4c_i32_i8), T(dot8c_i32_i4), T(dot2c_f32_f16),
                        T(pk_fma_f32), T(pk_mul_lo_u16), T(mad_i32_i16), T(max_i32_e32), T(min_u32), T(lshlrev_b32_v),
                        T(lshrrev_b32), T(ashr_sdwa), T(alignbit), T(cvt_pk_u8_f32), T(sat_pk_u8_i16), T(or_b32),
                        T(lshl_or_b32), T(sub_u32_sgpr), T(ds_read_b64), T(mix_mad_add), T(mix_salu_add), T(mix_med3_add),
                        T(mix_dsread_add)};
  const bool only_new = argc > 2 && !strcmp(argv[2], "new");
  const int first_new = 48;
  unsigned long long* d;
  CK(hipMalloc(&d, 32));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, %d iterations x 16 independent instructions per wave\n\n", prop.gcnArchName, cus, iters);
  printf("| instruction | cyc/inst/SIMD WPS=1 | WPS=2 | WPS=4 | WPS=8 | shader clock GHz (WPS=8) | wall ns/inst/SIMD (WPS=8) |\n|---|---|---|---|---|---|---|\n");
  int index = -1;
  for (const test& t : tests) {
    ++index;
    if (only_new && index < first_new) continue;
    double res[4], ghz = 0, wall = 0;
    int k = 0;
    for (int wps : {1, 2, 4, 8}) {
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(t.fn, dim3(cus * wps), dim3(256), 0, 0, d, 1000);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(t.fn, dim3(cus * wps), dim3(256), 0, 0, d, iters);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long h[4];
      CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
      res[k++] = (double)h[0] / ((double)iters * 16 * wps);
      ghz = (double)h[0] / ((double)h[2] * 10.0);      // s_memrealtime: 100 MHz
      wall = ms * 1e6 / ((double)iters * 16 * wps);
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("| %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.3f |\n", t.name, res[0], res[1], res[2], res[3], ghz, wall);
    fflush(stdout);
  }
  return 0;
}
