// Microbenchmark: cycles per wave64 VALU instruction on one SIMD at 1, 2, 4, 8 waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP256(x) REP16(REP16(x))

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters) {
  float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = 2.0f, e = 3.0f, f = 4.f, g = 5.f, h = 6.f;
  unsigned long long m0 = __ballot(threadIdx.x & 1), m1 = __ballot(threadIdx.x & 2);
  asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" ::"v"(a), "v"(b) : "vcc");
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {        // independent v_fma_f32 x8 streams
      REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
    } else if (KIND == 1) { // v_cndmask_b32 with VCC
      REP16(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c) : "vcc");)
    } else if (KIND == 2) { // v_cndmask_b32_e64 with SGPR-pair masks
      REP16(asm volatile("v_cndmask_b32_e64 %0, %4, %5, %6\n v_cndmask_b32_e64 %1, %4, %5, %7\n v_cndmask_b32_e64 %2, %4, %5, %6\n v_cndmask_b32_e64 %3, %4, %5, %7"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c), "s"(m0), "s"(m1));)
    } else if (KIND == 3) { // v_cmp_lt_u32_e64 to SGPR pair
      REP16(asm volatile("v_cmp_lt_u32_e64 %0, %2, %3\n v_cmp_lt_u32_e64 %1, %3, %2\n v_cmp_lt_u32_e64 %0, %2, %3\n v_cmp_lt_u32_e64 %1, %3, %2"
                         : "+s"(m0), "+s"(m1) : "v"(b), "v"(c));)
    } else if (KIND == 4) { // v_swap_b32
      REP16(asm volatile("v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %0, %1\n v_swap_b32 %2, %3"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f));)
    } else if (KIND == 5) { // dependent cmp -> cndmask pairs (select network level shape)
      REP16(asm volatile("v_cmp_lt_u32_e64 %4, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %2, %2, %3, %4\n v_cndmask_b32_e64 %3, %3, %2, %4"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f), "+s"(m0));)
    } else if (KIND == 6) { // v_min_u32 / v_max_u32
      REP16(asm volatile("v_min_u32 %0, %0, %4\n v_max_u32 %1, %1, %4\n v_min_u32 %2, %2, %5\n v_max_u32 %3, %3, %5"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
    } else if (KIND == 8) { // v_cndmask_b32_e64 with vcc named explicitly as the mask
      REP16(asm volatile("v_cndmask_b32_e64 %0, %4, %5, vcc\n v_cndmask_b32_e64 %1, %4, %5, vcc\n v_cndmask_b32_e64 %2, %4, %5, vcc\n v_cndmask_b32_e64 %3, %4, %5, vcc"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
    } else if (KIND == 9) { // e32 form, vcc written once before the loop by a real compare
      REP16(asm volatile("v_cndmask_b32_e32 %0, %4, %5, vcc\n v_cndmask_b32_e32 %1, %4, %5, vcc\n v_cndmask_b32_e32 %2, %4, %5, vcc\n v_cndmask_b32_e32 %3, %4, %5, vcc"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));)
    } else if (KIND == 10) { // v_cmp e32 -> vcc then 3 cndmask e32 (what a select bubble level looks like)
      REP16(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %2, vcc"
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) :: "vcc");)
    } else if (KIND == 7) { // v_pk_fma_f32 (2 fp32 per lane per instruction)
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2"
                         : "+v"(*(double*)&a), "+v"(*(double*)&e) : "v"(*(double*)&g));)
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = a + d + e + f + g + h + (float)(m0 ^ m1);
}

template <int KIND>
void run(const char* name) {
  const int iters = 200;
  printf("%-34s", name);
  for (int wps : {1, 2, 4, 8}) {            // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    const int blocks = 256 * wps;           // one block per CU per wave-per-SIMD
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, blocks * 256 * 4);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    const double per_wave = sum / h.size() / (iters * 64.0);    // cycles per instruction as seen by one wave
    printf("  %dw: %5.2f (SIMD: %4.2f)", wps, per_wave, per_wave / wps);
    hipFree(out); hipFree(sink);
  }
  printf("\n");
}

int main() {
  printf("cycles per wave64 instruction: per wave (and per SIMD = per wave / waves per SIMD)\n");
  run<0>("v_fma_f32 (4 independent)");
  run<1>("v_cndmask_b32 vcc");
  run<2>("v_cndmask_b32_e64 sgpr mask");
  run<3>("v_cmp_lt_u32_e64 -> sgpr");
  run<4>("v_swap_b32");
  run<5>("cmp -> 3 cndmask (dependent)");
  run<6>("v_min_u32 / v_max_u32");
  run<7>("v_pk_fma_f32");
  run<8>("v_cndmask_b32_e64 mask=vcc");
  run<9>("v_cndmask_b32_e32 vcc (no clobber)");
  run<10>("cmp_e32->vcc + 3 cndmask_e32");
  return 0;
}
