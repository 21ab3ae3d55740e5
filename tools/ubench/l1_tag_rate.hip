// Microbenchmark: what the vector L1 (TCP) of a gfx950 CU sustains for per-lane 16-byte gathers out of a table that
// stays L1-resident (16 KB), by how many distinct 128-byte lines a wave's 64 addresses touch.  Calibrates the "vmem-L1"
// roofline of bench.py (TCP_TOTAL_CACHE_ACCESSES per clock per CU): run it once plainly (cycles, s_memtime) and once under
//   rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE -- ./l1_tag_rate
// (accesses per dispatch; the kernel names carry the pattern).
// hipcc -O3 --offload-arch=gfx950 l1_tag_rate.hip -o l1_tag_rate && ./l1_tag_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kTableF4 = 1024;   // 16 KB of float4
constexpr int kIters = 2048, kUnroll = 8;

// KIND 0: every lane its own 128-B line (64 lines per instruction)
// KIND 1: 8 consecutive lanes share a line (8 lines per instruction: a coalesced 1 KB)
// KIND 2: all lanes one address (1 line)
// KIND 3: SH-like: 16 lanes name 6 distinct 192-B records at random, every lane reads 16 B at the same offset of its record
// KIND 4: 16 distinct lines per instruction (every quarter-wave 4 lines)
template <int KIND>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ tab, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  unsigned idx;
  if (KIND == 0) idx = lane * 8;
  else if (KIND == 1) idx = lane;
  else if (KIND == 2) idx = 0;
  else if (KIND == 3) idx = (((lane >> 4) * 6 + (lane * 2654435761u >> 28) % 6) * 37 % 80) * 12;
  else idx = (lane >> 2) * 8 + (lane & 3);
  idx = (idx + (threadIdx.x >> 6) * 64 + blockIdx.x * 16) % kTableF4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < kIters; ++i) {
    float4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) v[u] = tab[(idx + u * (KIND == 3 ? 1 : 128)) % kTableF4];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    idx = (idx + (KIND == 3 ? 12 * 7 : 8 * 13)) % kTableF4;
    asm volatile("" : "+v"(idx));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int KIND>
void run(const char* name, const float4* tab, unsigned long long* out, float* sink, int wgs_per_cu) {
  const int blocks = 256 * wgs_per_cu;
  hipLaunchKernelGGL(k_gather<KIND>, dim3(blocks), dim3(256), 0, 0, tab, out, sink);   // warm
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_gather<KIND>, dim3(blocks), dim3(256), 0, 0, tab, out, sink);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cyc = (double)h[h.size() / 2];   // a wave's loop, shader cycles (all waves of a CU run side by side)
  const double loads_per_cu = (double)wgs_per_cu * 4 * kIters * kUnroll;
  printf("%-44s %2d waves/CU: %7.3f ms  wave loop %9.0f cycles  -> %.4f gather instructions / clk / CU (%.2f clk each)\n", name,
         wgs_per_cu * 4, ms, cyc, loads_per_cu / cyc, cyc / loads_per_cu);
}

int main() {
  float4* tab; unsigned long long* out; float* sink;
  hipMalloc(&tab, kTableF4 * 16); hipMalloc(&out, 256 * 8 * 4 * 8); hipMalloc(&sink, 256 * 8 * 256 * 4);
  hipMemset(tab, 0, kTableF4 * 16);
  for (int w : {2, 4}) {
    run<0>("64 lines / instruction (lane = line)", tab, out, sink, w);
    run<4>("16 lines / instruction", tab, out, sink, w);
    run<1>("8 lines / instruction (coalesced 1 KB)", tab, out, sink, w);
    run<2>("1 line / instruction (broadcast)", tab, out, sink, w);
    run<3>("SH-like: 6 records per 16 lanes", tab, out, sink, w);
  }
  return 0;
}
