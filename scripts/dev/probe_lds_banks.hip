// Probe: LDS array cycles per wave-instruction for the access patterns of the fused MLP backward, for candidate tile
// layouts.  16 waves (four per SIMD) hammer the LDS with the same pattern; cycles per instruction per wave = LDS cost when
// the array is the bottleneck.  Validates scripts/dev/lds_sim.py against the hardware (incl. ds_read_b64_tr_b16).
//   hipcc --offload-arch=gfx950 -O3 -o probe_lds_banks probe_lds_banks.hip && ./probe_lds_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <functional>
typedef short s4 __attribute__((ext_vector_type(4)));
enum { OP_TR = 0, OP_RD64 = 1, OP_WR64 = 2, OP_RD128 = 3 };
template <int OP>
__global__ __launch_bounds__(1024) void k(const uint32_t* addr, int iters, long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) ((uint32_t*)lds)[i] = i;
  __syncthreads();
  const uint32_t a = addr[threadIdx.x & 63];
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      if (OP == OP_TR) { s4 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a)); }
      else if (OP == OP_RD64) { uint2 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); }
      else if (OP == OP_WR64) { uint2 v = make_uint2(1u, 2u); asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(v)); }
      else { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
static double g_ticks_per_ns = 0;
typedef std::function<int(int, int)> Layout;   // (row, col_halves) -> halves offset
static double run(int op, const Layout& f, std::function<void(int, int&, int&)> rc) {
  uint32_t h[64];
  for (int l = 0; l < 64; ++l) { int r, c; rc(l, r, c); h[l] = (uint32_t)f(r, c) * 2; }
  static uint32_t* d = nullptr; static long long* o = nullptr;
  if (!d) { hipMalloc(&d, 256); hipMalloc(&o, 16); }
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  const int iters = 20000;
  static hipEvent_t e0, e1; static bool ev = false;
  if (!ev) { hipEventCreate(&e0); hipEventCreate(&e1); ev = true; }
  auto launch = [&](auto kern) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 160 * 1024, 0, d, iters, o);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 160 * 1024, 0, d, iters, o);
    hipEventRecord(e1, 0);
  };
  if (op == OP_TR) launch(k<OP_TR>); else if (op == OP_RD64) launch(k<OP_RD64>); else if (op == OP_WR64) launch(k<OP_WR64>); else launch(k<OP_RD128>);
  hipDeviceSynchronize();
  long long t; hipMemcpy(&t, o, 8, hipMemcpyDeviceToHost);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  g_ticks_per_ns = (double)t / (ms * 1e6);
  return (double)t / (iters * 12) / 16.0;   // LDS cycles per wave-instruction with 16 waves saturating the array
}
int main() {
  struct { const char* name; Layout f; } layouts[] = {
    {"pitch136", [](int r, int c) { return r * 136 + c; }},
    {"pitch132", [](int r, int c) { return r * 132 + c; }},
    {"pitch160", [](int r, int c) { return r * 160 + c; }},
    {"pitch128", [](int r, int c) { return r * 128 + c; }},
    {"swz128", [](int r, int c) { int d = c >> 1; int p = ((d ^ (16 * (r & 3))) + 2 * ((r >> 2) & 7)) & 63; return r * 128 + 2 * p + (c & 1); }},
    {"swz128_rot4", [](int r, int c) { int d = c >> 1; int p = ((d ^ (16 * (r & 3))) + 4 * ((r >> 2) & 3)) & 63; return r * 128 + 2 * p + (c & 1); }},
  };
  for (auto& L : layouts) {
    double tr = run(OP_TR, L.f, [](int l, int& r, int& c) { r = 8 * (l >> 5) + ((l & 15) >> 2); c = 32 + 16 * ((l >> 4) & 1) + 4 * (l & 3); });
    double trc = run(OP_TR, L.f, [](int l, int& r, int& c) { r = 16 + 4 * (l >> 5) + ((l & 15) >> 2) + 8; c = 64 + 16 * ((l >> 4) & 1) + 4 * (l & 3); });
    double rd = run(OP_RD64, L.f, [](int l, int& r, int& c) { r = 32 + (l & 31); c = 32 + 8 + 4 * (l >> 5); });
    double wr = run(OP_WR64, L.f, [](int l, int& r, int& c) { r = 32 + (l & 31); c = 32 + 8 + 4 * (l >> 5); });
    double r128 = run(OP_RD128, L.f, [](int l, int& r, int& c) { r = 32 + (l & 31); c = 16 + 8 * (l >> 5); });
    printf("[%.2f ticks/ns incl. launch] %-12s tr_nat %.2f  tr_chain %.2f  rd_b64_row %.2f  wr_b64_row %.2f  rd_b128_row %.2f\n", g_ticks_per_ns, L.name, tr, trc, rd, wr, r128);
  }
  return 0;
}
