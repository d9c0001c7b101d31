// Probe: bank-conflict share (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE) and cycles per access of every LDS access pattern of
// k_mlp_bwd128 (mlp_bwd128.hip), one kernel per pattern, one wave per SIMD as in the product kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int PH = 148;
__device__ inline int prow(int f) { return (f & ~31) | ((f & 3) << 3) | ((f >> 2) & 7); }
// KIND: 0 tr-read (b64_tr_b16), 1 ds_read_b64, 2 ds_read_b128, 3 ds_write_b64
template <int PAT>
__device__ inline void addrs(int lane, int wave, uint32_t (&a)[16], int& kind) {
  const int hf = lane >> 5, c = lane & 31;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    int off = 0;   // halves
    if (PAT == 0) { kind = 3; const int rb = s & 3, q = (s >> 2) & 3; off = prow(32 * wave + c) * PH + 32 * rb + 4 * hf + 8 * q; }                       // write_slice
    if (PAT == 1) { kind = 1; const int ib = s & 3, kd = (s >> 2); off = prow(32 * ib + c) * PH + 32 * (kd >> 1) + 16 * (kd & 1) + 4 * hf + 8 * (s & 1); } // samp_frag (p0 / p1 alternate)
    if (PAT == 2) { kind = 0; const int ks = s & 7, ib = s >> 3; const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);           // bx: tr_frag_chained on x, pitch 56
                    off = row * 56 + 32 * ib + 16 * ((lane >> 4) & 1) + 4 * (lane & 3); }
    if (PAT == 3) { kind = 0; const int ks = s & 7; const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);                        // bx, pitch 40 (colour head)
                    off = row * 40 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3); }
    if (PAT == 4) { kind = 0; const int ks = s & 7; const int row = 32 * (ks >> 1) + 16 * (ks & 1) + 4 * hf + ((lane & 15) >> 2);                        // fo: tr_frag_chained on dOut, pitch 16
                    off = row * 16 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3); if (off > 128 * 16 - 4) off = 0; }
    if (PAT == 5) { kind = 2; const int rb = s & 3, ks = (s >> 2) % 3; off = (32 * rb + c) * 56 + 16 * ks + 8 * hf; }                                     // x row reads (layer 0), pitch 56
    if (PAT == 6) { kind = 3; const int ib = s & 1, q = (s >> 1) & 3; off = wave * 32 * 48 + c * 48 + 32 * ib + 8 * q + 4 * hf; if (32 * ib + 8 * q + 4 * hf >= 48) off = wave * 32 * 48 + c * 48; }   // d_in stage writes, pitch 48
    if (PAT == 7) { kind = 2; off = (32 * (s & 3) + c) * 16 + 8 * hf; }                                                                                   // ao: dOut rows, pitch 16
    if (PAT == 8) { kind = 1; const int rb = s & 3, q = (s >> 2) & 3; off = prow(32 * wave + c) * PH + 32 * rb + 4 * hf + 8 * q; }                       // req_m (mask reads)
    if (PAT == 9) { kind = 2; off = ((s * 64) + lane) * 8; }
    if (PAT == 10) { kind = 3; off = ((s * 64) + lane) * 4; }                                                                                             // ds_write_b64, linear
    if (PAT == 11) { kind = 4; off = ((s * 64) + lane) * 8; }                                                                                             // ds_write_b128, linear
    if (PAT == 12) { kind = 4; const int rb = s & 3, q2 = (s >> 2) & 1; off = prow(32 * wave + c) * PH + 32 * rb + 16 * q2 + 8 * hf; }                   // 16-byte slice stores (permuted sample order)
    if (PAT == 13) { kind = 3; const int rb = s & 3, q = (s >> 2) & 3; off = prow(32 * wave + c) * PH + 32 * rb + 16 * (q >> 1) + 8 * hf + 4 * (q & 1); } // 8-byte stores, halves 8 apart                                                                                              // w0t fragments (linear)
    a[s] = (uint32_t)(off * 2);
  }
}
template <int PAT>
__global__ __launch_bounds__(256) void k_pat(long long* out, int rounds) {
  extern __shared__ __attribute__((aligned(16))) short lds[];
  for (int i = threadIdx.x; i < 70 * 1024; i += 256) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t a[16]; int kind = 0;
  addrs<PAT>(lane, wave, a, kind);
  u4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    if (kind == 0) { s4 v[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[s]) : "v"(a[s]));
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int s = 0; s < 16; ++s) { acc[0] ^= (uint32_t)v[s][0]; acc[1] ^= (uint32_t)v[s][3]; }
    } else if (kind == 1) { u2 v[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("ds_read_b64 %0, %1" : "=v"(v[s]) : "v"(a[s]));
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int s = 0; s < 16; ++s) { acc[0] ^= v[s][0]; acc[1] ^= v[s][1]; }
    } else if (kind == 2) { u4 v[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(v[s]) : "v"(a[s]));
      asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
      for (int s = 0; s < 16; ++s) acc ^= v[s];
    } else if (kind == 3) { const u2 w = {(uint32_t)r, acc[0]};
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("ds_write_b64 %0, %1" :: "v"(a[s]), "v"(w) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)");
    } else { const u4 w = {(uint32_t)r, acc[0], acc[1], 7u};
#pragma unroll
      for (int s = 0; s < 16; ++s) asm volatile("ds_write_b128 %0, %1" :: "v"(a[s]), "v"(w) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) out[1] = acc[2];
}
template <int PAT> void run(const char* name, long long* d) {
  const int rounds = 2000;
  (void)hipFuncSetAttribute((const void*)k_pat<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_pat<PAT>, dim3(256), dim3(256), 150 * 1024, 0, d, rounds);
  (void)hipDeviceSynchronize();
  long long h[2]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("k_pat<%d> %-52s %7.1f ticks per wave access (16 in flight)\n", PAT, name, (double)h[0] / (rounds * 16.0));
}
int main() {
  long long* d; (void)hipMalloc(&d, 16);
  run<0>("write_slice (ds_write_b64, prow rows, pitch 148)", d);
  run<1>("samp_frag (ds_read_b64, prow rows)", d);
  run<2>("bx: transposed x reads, pitch 56 (density head)", d);
  run<3>("bx: transposed x reads, pitch 40 (colour head)", d);
  run<4>("fo: transposed dOut reads, pitch 16", d);
  run<5>("x rows for layer 0 (ds_read_b128, pitch 56)", d);
  run<6>("d_in stage writes (ds_write_b64, pitch 48)", d);
  run<7>("ao: dOut rows (ds_read_b128, pitch 16)", d);
  run<8>("req_m mask reads (ds_read_b64, prow rows)", d);
  run<9>("w0t fragments (ds_read_b128, linear)", d);
  run<10>("ds_write_b64 linear", d);
  run<11>("ds_write_b128 linear", d);
  run<12>("16-byte slice stores, prow rows (sample order permuted)", d);
  run<13>("8-byte slice stores, halves 8 apart", d);
  return 0;
}
