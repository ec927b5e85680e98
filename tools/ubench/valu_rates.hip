// Development tool: issue cost of single VALU instructions on gfx950, one wave per SIMD and two, from a loop of 64 independent
// instances per trip (8 accumulators).  hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int trips, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3, f4 = (float)a4, f5 = (float)a5, f6 = (float)a6, f7 = (float)a7;
  int i0 = threadIdx.x;
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) {
#define X(n) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 1) {
#define X(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 2) {
#define X(n) asm volatile("v_trunc_f64 %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 3) {
#define X(n) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 4) {
#define X(n) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 5) {
#define X(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 6) {
#define X(n) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f##n) : "v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 7) {
#define X(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##n) : "v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 8) {
#define X(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i0) : "v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 9) {
#define X(n) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 10) {
#define X(n) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 11) {   // dependent chain: one accumulator
#define X(n) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a0));
        REP8(X)
#undef X
      } else if (OP == 12) {   // dependent add chain
#define X(n) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a0));
        REP8(X)
#undef X
      } else if (OP == 13) {   // fma + a scalar instruction between (does SALU take a VALU slot of the same wave?)
#define X(n) asm volatile("v_fma_f64 %0, %0, %0, %0\n\ts_mov_b32 s20, 1" : "+v"(a##n) : : "s20");
        REP8(X)
#undef X
      } else if (OP == 14) {   // v_ldexp / v_cvt_f64_i32
#define X(n) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a##n) : "v"(i0));
        REP8(X)
#undef X
      } else if (OP == 16) {   // select through an SGPR-pair mask
#define X(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f##n) : "v"(f0) : "s20", "s21");
        REP8(X)
#undef X
      } else if (OP == 17) {   // select, distinct sources
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f3) : "v"(f4), "v"(f5));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f6) : "v"(f7), "v"(f1));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f3) : "v"(f4), "v"(f5));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f6) : "v"(f7), "v"(f1));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f0) : "v"(f1), "v"(f2));
        asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f3) : "v"(f4), "v"(f5));
      } else if (OP == 18) {   // compare writing vcc
#define X(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a##n), "v"(a0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 19) {
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(f##n), "v"(f0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 20) {
#define X(n) asm volatile("v_mov_b32 %0, %0" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 21) {
#define X(n) asm volatile("v_add_u32 %0, %0, %0" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 22) {
#define X(n) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(f##n) : "s20");
        REP8(X)
#undef X
      } else if (OP == 23) {
#define X(n) asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(f##n));
        REP8(X)
#undef X
      } else if (OP == 24) {   // compare into an SGPR pair + select from it (the usual pattern)
#define X(n) asm volatile("v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f##n) : "v"(f0) : "s20", "s21");
        REP8(X)
#undef X
      } else if (OP == 25) {   // 64-bit move
#define X(n) asm volatile("v_mov_b64 %0, %0" : "+v"(a##n));
        REP8(X)
#undef X
      } else if (OP == 26) {   // select with a mask register: bfi
#define X(n) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(f##n) : "v"(i0), "v"(f0));
        REP8(X)
#undef X
      } else if (OP == 27) {   // scalar ALU alone
#define X(n) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
        REP8(X)
#undef X
      } else if (OP == 28) {   // s_and_saveexec + restore
#define X(n) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]" : : : "s20", "s21", "scc");
        REP8(X)
#undef X
      } else if (OP == 29) {   // fma with exec toggled around (does writing exec stall the VALU?)
#define X(n) asm volatile("s_mov_b64 s[20:21], exec\n\tv_fma_f64 %0, %0, %0, %0\n\ts_mov_b64 exec, s[20:21]" : "+v"(a##n) : : "s20", "s21");
        REP8(X)
#undef X
      } else if (OP == 30) {   // the usual pair: compare into vcc, VOP2 select from vcc
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f##n) : "v"(f0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 31) {   // compare into vcc, two VOP2 selects (a 64-bit select)
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cndmask_b32 %2, %2, %1, vcc" : "+v"(f##n) : "v"(f0), "v"(i0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 32) {   // VOP3 encoding of the select, mask still vcc
#define X(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(f##n) : "v"(f0));
        REP8(X)
#undef X
      } else if (OP == 33) {   // compare into vcc, VOP3-encoded select from vcc
#define X(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(f##n) : "v"(f0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 34) {   // add with carry-out/in through vcc (the other implicit vcc reader)
#define X(n) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(f##n) : "v"(f0) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 15) {   // v_bfi_b32
#define X(n) asm volatile("v_bfi_b32 %0, %0, %0, %0" : "+v"(f##n));
        REP8(X)
#undef X
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0;
}

template <int OP>
double run(int waves_per_simd, const char *name) {
  int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * waves_per_simd;      // 256 threads = 4 waves = one per SIMD
  double *out; hipMalloc(&out, sizeof(double) * blocks * 256);
  const int trips = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 100, 1.5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, trips, 1.5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_wave = (double)trips * 64;
  const double ns_per_inst = ms * 1e6 / inst_per_wave;       // per wave-instruction of ONE wave (waves_per_simd run concurrently)
  printf("%-28s waves/SIMD %d: %.3f ns per instruction per wave -> %.2f cycles @2.4GHz per SIMD-instruction\n", name, waves_per_simd, ns_per_inst, ns_per_inst * 2.4 / waves_per_simd);
  hipFree(out);
  return ns_per_inst;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, "v_fma_f64"); run<3>(w, "v_add_f64"); run<4>(w, "v_mul_f64"); run<1>(w, "v_rcp_f64"); run<2>(w, "v_trunc_f64");
    run<8>(w, "v_cvt_i32_f64"); run<14>(w, "v_cvt_f64_i32"); run<6>(w, "v_cvt_f32_f64"); run<7>(w, "v_cvt_f64_f32"); run<5>(w, "v_rcp_f32");
    run<9>(w, "v_fma_f32"); run<10>(w, "v_cndmask_b32"); run<15>(w, "v_bfi_b32"); run<11>(w, "v_fma_f64 dependent chain"); run<12>(w, "v_add_f64 dependent chain");
    run<13>(w, "v_fma_f64 + s_mov");
    run<16>(w, "v_cndmask_b32_e64 (sgpr mask)"); run<17>(w, "v_cndmask_b32 distinct regs"); run<18>(w, "v_cmp_lt_f64 -> vcc"); run<19>(w, "v_cmp_lt_u32 -> vcc");
    run<20>(w, "v_mov_b32"); run<21>(w, "v_add_u32"); run<22>(w, "v_readlane_b32"); run<23>(w, "v_mov_b32_dpp row_mirror"); run<24>(w, "v_cmp_e64 + v_cndmask_e64");
    run<30>(w, "v_cmp vcc + v_cndmask_e32"); run<31>(w, "v_cmp vcc + 2 v_cndmask_e32"); run<32>(w, "v_cndmask_b32_e64 .., vcc"); run<33>(w, "v_cmp vcc + v_cndmask_e64 vcc"); run<34>(w, "v_addc_co_u32 vcc");
    run<25>(w, "v_mov_b64"); run<26>(w, "v_bfi_b32 (mask select)"); run<27>(w, "s_add_u32"); run<28>(w, "s_and_saveexec + s_or exec"); run<29>(w, "fma between exec save/restore");
  }
  return 0;
}
