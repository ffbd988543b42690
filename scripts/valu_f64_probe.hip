// Issue cost and dependent latency of the fp64 vector instructions the register Cholesky is made
// of, one wave alone on a SIMD (hipcc --offload-arch=gfx950 -O3 scripts/valu_f64_probe.hip).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define PROBE(name, setup, body)                                                  \
  __global__ void name(double* p, unsigned long long* out) {                       \
    double a = p[threadIdx.x], b = p[64 + threadIdx.x], c = p[128 + threadIdx.x], d = p[192 + threadIdx.x]; \
    double e = a + 1.0, f = b + 2.0, g = c + 3.0, h = d + 4.0;                    \
    setup;                                                                         \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
    for (int it = 0; it < 16; ++it) { REP64(body) }                               \
    asm volatile("s_nop 0" ::: "memory");                                         \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                            \
    p[threadIdx.x] = a + b + c + d + e + f + g + h;                               \
    if (threadIdx.x == 0) out[0] = t1 - t0;                                       \
  }
PROBE(k_fma_dep, , asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
PROBE(k_fma_ind, , asm volatile("v_fma_f64 %0, %4, %5, %0\n v_fma_f64 %1, %4, %5, %1\n v_fma_f64 %2, %4, %5, %2\n v_fma_f64 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
PROBE(k_fmac_dpp_ind, , asm volatile("v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
PROBE(k_fmac_dpp_dep, , asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(e), "v"(f));)
PROBE(k_fmac_e32_ind, , asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
PROBE(k_mov64_dpp_ind, , asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
PROBE(k_mov32_dpp_ind, int x0 = (int)threadIdx.x; int x1 = x0 + 1; int x2 = x0 + 2; int x3 = x0 + 3; int y0 = x0 * 3, asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y0));)
PROBE(k_rcp_dep, , asm volatile("v_rcp_f64 %0, %0\n s_nop 0" : "+v"(a));)
PROBE(k_rcp_ind, , asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %5\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
PROBE(k_rsq_dep, , asm volatile("v_rsq_f64 %0, %0\n s_nop 0" : "+v"(a));)
PROBE(k_mul_dep, , asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));)
PROBE(k_fma32_dep, float fa = (float)a; float fb = (float)b, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa) : "v"(fb));)
PROBE(k_dppmov_then_fma, , asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_mov_b64_dpp %3, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %3, %1, %2" : "+v"(a), "+v"(e), "+v"(f), "+v"(g));)
PROBE(k_readlane_pair, int lo = 0; int hi = 0, asm volatile("v_readlane_b32 %0, %2, 3\n v_readlane_b32 %1, %3, 3\n" : "=s"(lo), "=s"(hi) : "v"(__double2loint(a)), "v"(__double2hiint(a)));)
struct P { const char* name; void (*k)(double*, unsigned long long*); int per; };
int main() {
  double* d; unsigned long long* o;
  hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&o, 64);
  P ps[] = {{"v_fma_f64 dependent", k_fma_dep, 1}, {"v_fma_f64 independent (4)", k_fma_ind, 4},
            {"v_fmac_f64 e32 independent (4)", k_fmac_e32_ind, 4},
            {"v_fmac_f64_dpp independent (4)", k_fmac_dpp_ind, 4}, {"v_fmac_f64_dpp dependent", k_fmac_dpp_dep, 1},
            {"v_mov_b64_dpp independent (4)", k_mov64_dpp_ind, 4}, {"v_mov_b32_dpp independent (4)", k_mov32_dpp_ind, 4},
            {"v_rcp_f64 dependent (+s_nop 0)", k_rcp_dep, 1}, {"v_rcp_f64 independent (4)", k_rcp_ind, 4},
            {"v_rsq_f64 dependent (+s_nop 0)", k_rsq_dep, 1}, {"v_mul_f64 dependent", k_mul_dep, 1},
            {"v_fma_f32 dependent", k_fma32_dep, 1},
            {"fmac_dpp; s_nop 1; mov_b64_dpp; fma (pivot hand-over)", k_dppmov_then_fma, 1},
            {"v_readlane_b32 x2", k_readlane_pair, 1}};
  for (auto& p : ps) {
    unsigned long long best = ~0ull;
    for (int r = 0; r < 5; ++r) {
      hipLaunchKernelGGL(p.k, dim3(1), dim3(64), 0, 0, d, o);
      unsigned long long v; hipMemcpy(&v, o, 8, hipMemcpyDeviceToHost);
      if (v < best) best = v;
    }
    printf("%-58s %7.2f cycles per instruction (s_memtime ticks / %d)\n", p.name, (double)best / (16.0 * 64 * p.per), 16 * 64 * p.per);
  }
  return 0;
}
