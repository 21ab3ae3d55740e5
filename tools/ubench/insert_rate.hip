// Microbenchmark: the chain-free sorted-list insertion of k_trace (16 compares + 64 selects on two register
// arrays) in isolation, at 1..8 waves per SIMD: cycles per insertion and per VALU instruction.
// hipcc -O3 --offload-arch=gfx950 insert_rate.hip -o insert_rate && ./insert_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kLo, int kHi, int kK>
__device__ __forceinline__ void list_insert2(unsigned (&kt)[kK], unsigned (&kp)[kK], unsigned tn, unsigned pn) {
  constexpr int kB = kLo > 0 ? kLo - 1 : 0;
  bool lt[kHi - kB];
#pragma unroll
  for (int k = kB; k < kHi; ++k) lt[k - kB] = tn < kt[k];
#pragma unroll
  for (int k = kHi - 1; k > kB; --k) {
    kt[k] = lt[k - 1 - kB] ? kt[k - 1] : (lt[k - kB] ? tn : kt[k]);
    kp[k] = lt[k - 1 - kB] ? kp[k - 1] : (lt[k - kB] ? pn : kp[k]);
  }
  if constexpr (kLo == 0) {
    kt[0] = lt[0] ? tn : kt[0];
    kp[0] = lt[0] ? pn : kp[0];
  }
}

// The same update written out: compares into SGPR pairs (never VCC), selects in the e64 form reading them.
// Slots hi .. hi-3 from the ORIGINAL values of slots hi-4 .. hi (blocks run from the top slot down).
#define INS4(K3, K2, K1, K0, NB)                                                                         \
  asm volatile(                                                                                          \
      "v_cmp_lt_u32_e64 %[m0], %[tn], %[t3]\n\t"                                                         \
      "v_cmp_lt_u32_e64 %[m1], %[tn], %[t2]\n\t"                                                         \
      "v_cmp_lt_u32_e64 %[m2], %[tn], %[t1]\n\t"                                                         \
      "v_cmp_lt_u32_e64 %[m3], %[tn], %[t0]\n\t"                                                         \
      "v_cmp_lt_u32_e64 %[m4], %[tn], %[tb]\n\t"                                                         \
      "v_cndmask_b32_e64 %[t3], %[t3], %[tn], %[m0]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p3], %[p3], %[pn], %[m0]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t3], %[t3], %[t2], %[m1]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p3], %[p3], %[p2], %[m1]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t2], %[t2], %[tn], %[m1]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p2], %[p2], %[pn], %[m1]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t2], %[t2], %[t1], %[m2]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p2], %[p2], %[p1], %[m2]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t1], %[t1], %[tn], %[m2]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p1], %[p1], %[pn], %[m2]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t1], %[t1], %[t0], %[m3]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p1], %[p1], %[p0], %[m3]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t0], %[t0], %[tn], %[m3]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p0], %[p0], %[pn], %[m3]\n\t"                                                 \
      "v_cndmask_b32_e64 %[t0], %[t0], %[tb], %[m4]\n\t"                                                 \
      "v_cndmask_b32_e64 %[p0], %[p0], %[pb], %[m4]"                                                      \
      : [t3] "+v"(kt[K3]), [t2] "+v"(kt[K2]), [t1] "+v"(kt[K1]), [t0] "+v"(kt[K0]), [p3] "+v"(kp[K3]),   \
        [p2] "+v"(kp[K2]), [p1] "+v"(kp[K1]), [p0] "+v"(kp[K0]), [m0] "=&s"(m0), [m1] "=&s"(m1),         \
        [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4)                                                   \
      : [tn] "v"(tn), [pn] "v"(pn), [tb] "v"(kt[NB]), [pb] "v"(kp[NB]))

__global__ __launch_bounds__(256) void k_asm(unsigned long long* out, unsigned* sink, int iters) {
  extern __shared__ char lds[];
  unsigned kt[17], kp[17];   // [0] is a sentinel below slot 1: 0 is never > tn, so slot 1's "left neighbour" select is off
#pragma unroll
  for (int i = 0; i < 17; ++i) { kt[i] = i == 0 ? 0u : 0x7F800000u; kp[i] = i; }
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const unsigned tn = x >> 2, pn = (unsigned)i << 4;
    unsigned long long m0, m1, m2, m3, m4;
    INS4(16, 15, 14, 13, 12);
    INS4(12, 11, 10, 9, 8);
    INS4(8, 7, 6, 5, 4);
    INS4(4, 3, 2, 1, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 17; ++i) s += kt[i] ^ kp[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s + (lds[threadIdx.x] & 0);
}

template <int kSlots>
__global__ __launch_bounds__(256) void k(unsigned long long* out, unsigned* sink, int iters) {
  extern __shared__ char lds[];   // only to cap the number of resident workgroups
  unsigned kt[16], kp[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { kt[i] = 0x7F800000u; kp[i] = i; }
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;             // 1 VALU (mad)
    const unsigned tn = x >> 2;                  // 1 VALU
    list_insert2<0, kSlots>(kt, kp, tn, (unsigned)i << 4);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += kt[i] ^ kp[i];
  sink[blockIdx.x * 256 + threadIdx.x] = s + (lds[threadIdx.x] & 0);
}

template <typename F>
void run_f(F kern, const char* name, int valu_per_iter);

template <int kSlots>
void run(const char* name, int valu_per_iter) { run_f(k<kSlots>, name, valu_per_iter); }

template <typename F>
void run_f(F kern, const char* name, int valu_per_iter) {
  const int iters = 2000;
  printf("%-28s", name);
  for (int waves : {1, 2, 4, 5, 6, 7, 8}) {
    const int blocks = 256 * waves;                       // `waves` workgroups of 4 waves per CU = `waves` per SIMD
    const size_t lds = (160 * 1024 / waves) & ~1023u;     // so that no more than `waves` are resident per CU
    unsigned long long* out; unsigned* sink;
    hipMalloc(&out, blocks * 4 * 8); hipMalloc(&sink, blocks * 256 * 4);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds > 65536 ? 65536 : lds, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double per_iter = sum / h.size() / iters;       // cycles one wave spends per insertion
    printf("  %dw: %6.1f cyc/ins = %4.2f cyc/VALU/SIMD", waves, per_iter, per_iter / waves / valu_per_iter);
    hipFree(out); hipFree(sink);
  }
  printf("\n");
}

int main() {
  run<16>("insert, 16 slots (82 VALU)", 82);
  run<8>("insert, 8 slots (42 VALU)", 42);
  run_f(k_asm, "asm e64, 16 slots (86 VALU)", 86);
  return 0;
}
