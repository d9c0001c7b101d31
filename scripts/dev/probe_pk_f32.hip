// v_pk_mul_f32 on gfx950 next to a second process.  scripts/dev/stress_scatter.py traced the run-to-run differences of the binned
// scatter (k_encode_bwd_bin) to lanes 48..63 of single waves whose PACKED fp32 products (interpolation weight x gradient, emitted by
// hipcc as v_pk_mul_f32 with op_sel operand picks) came out as signed zeros -- only while another process keeps the GPU busy; the same
// kernel compiled without packed fp32 instructions (-Xclang -target-feature -Xclang -packed-fp32-ops) is bit-stable.
// This probe multiplies the same pairs with v_pk_mul_f32 (four operand-select forms) and with scalar v_mul_f32 and reports every
// disagreement: which form, which lanes, and on which XCC / SE / CU / SIMD the wave ran (a single defective unit or all of them?).
//   hipcc --offload-arch=gfx950 -O3 -o probe_pk_f32 probe_pk_f32.hip ; ./probe_pk_f32 [seconds] [role: probe | load]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <chrono>
typedef float f2 __attribute__((ext_vector_type(2)));

struct Event { uint32_t lane, hwid, xcc, form, w, g, pk, scalar; };

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
// SRC: where the packed multiplies' second factor comes from -- 0: plain fp32 arithmetic, 1: a packed fp16 word unpacked by
// v_cvt_f32_f16 right in front of them (the gradient word of the scatter), 2: v_floor_f32 / v_sub (its interpolation weights)
template <int SRC, bool OTHER = false>
__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ xs, uint32_t* __restrict__ counts, Event* __restrict__ ev, int iters) {
  // counts: [0] wrong results, [1] events logged, [2] waves run, [8 .. 8+64) by lane, [72 .. 76) by form, [128 .. 128+4096) by unit
  __shared__ uint32_t tab[12288];   // 48 KB: the occupancy of k_encode_bwd_bin
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 12288; i += 512) tab[i] = (uint32_t)i * 2654435761u >> 20;
  __syncthreads();
  float x = xs[blockIdx.x * 512 + tid], y = xs[(blockIdx.x * 512 + tid) ^ 1] * 1.25f;
  uint32_t hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const uint32_t unit = ((xcc & 7u) << 9) | (((hwid >> 13) & 7u) << 6) | (((hwid >> 8) & 15u) << 2) | ((hwid >> 4) & 3u);
  if (lane == 0) atomicAdd(&counts[2], 1u);
  uint32_t nbad = 0;
  for (int it = 0; it < iters; ++it) {
    f2 g = {x, y};
    if (SRC == 1) { h16x2 gg; gg[0] = (h16)x; gg[1] = (h16)y; uint32_t gw = *(const uint32_t*)&gg; asm volatile("" : "+v"(gw)); gg = *(const h16x2*)&gw; g = f2{(float)gg[0], (float)gg[1]}; }
    if (SRC == 2) { const f2 sp = {x * 37.f + 0.5f, y * 53.f + 0.5f}; g = sp - f2{floorf(sp.x), floorf(sp.y)}; }
    const f2 w = {fabsf(y) * 0.5f + 0.125f, fabsf(x) * 0.25f + 0.0625f};
    f2 p0 = w * g, p1 = w * g.yx, p2 = w.xx * g, p3 = w.yy * g.yx;
    float q[8];
    const float ws[8] = {w.x, w.y, w.x, w.y, w.x, w.x, w.y, w.y}, gs[8] = {g.x, g.y, g.y, g.x, g.x, g.y, g.y, g.x};
    if (OTHER) {
      // the other packed families with the same crossed operand pick: v_pk_add_f32, v_pk_fma_f32 (fp32) and v_pk_mul_f16 / v_pk_fma_f16
      // (the 16-bit ones select halves of ONE register through op_sel; their results are compared as fp32 bit patterns of the halves)
      h16x2 hw, hg; hw[0] = (h16)w.x; hw[1] = (h16)w.y; hg[0] = (h16)g.x; hg[1] = (h16)g.y;
      uint32_t hgw = *(const uint32_t*)&hg; asm volatile("v_add_u32_e32 %0, 0, %0" : "+v"(hgw)); hg = *(const h16x2*)&hgw;   // freshly written
      h16x2 m16, f16v;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(p0) : "v"(w), "v"(g));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(p1) : "v"(w), "v"(g));
      asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(m16) : "v"(hw), "v"(hg));
      asm volatile("v_pk_fma_f16 %0, %1, %2, %1 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(f16v) : "v"(hw), "v"(hg));
      p2 = f2{(float)m16[0], (float)m16[1]}; p3 = f2{(float)f16v[0], (float)f16v[1]};
      asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(q[0]) : "v"(w.x), "v"(g.y));
      asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(q[1]) : "v"(w.y), "v"(g.x));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(q[2]) : "v"(w.x), "v"(g.y));
      asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(q[3]) : "v"(w.y), "v"(g.x));
      h16 r16[4]; const h16 a0 = hw[0], a1 = hw[1], b0 = hg[0], b1 = hg[1];
      asm volatile("v_mul_f16_e32 %0, %1, %2" : "=v"(r16[0]) : "v"(a0), "v"(b1));
      asm volatile("v_mul_f16_e32 %0, %1, %2" : "=v"(r16[1]) : "v"(a1), "v"(b0));
      asm volatile("v_fma_f16 %0, %1, %2, %1" : "=v"(r16[2]) : "v"(a0), "v"(b1));
      asm volatile("v_fma_f16 %0, %1, %2, %1" : "=v"(r16[3]) : "v"(a1), "v"(b0));
#pragma unroll
      for (int k = 0; k < 4; ++k) q[4 + k] = (float)r16[k];
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(q[k]) : "v"(ws[k]), "v"(gs[k]));
    }
    const float ps[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (__float_as_uint(ps[k]) != __float_as_uint(q[k])) {
        ++nbad;
        atomicAdd(&counts[8 + lane], 1u); atomicAdd(&counts[72 + (k >> 1)], 1u); atomicAdd(&counts[128 + unit], 1u);
        const uint32_t e = atomicAdd(&counts[1], 1u);
        if (e < 64) ev[e] = Event{(uint32_t)lane, hwid, xcc, (uint32_t)k, __float_as_uint(ws[k]), __float_as_uint(gs[k]), __float_as_uint(ps[k]), __float_as_uint(q[k])};
      }
    x = x * 1.0009765625f + (float)(tab[(tid + it) % 12288] & 3u) * 1e-5f + ps[3] * 1e-3f;
    y = y * 0.9990234375f - 1e-4f + ps[6] * 1e-3f;
    if (!(fabsf(x) < 1e4f)) x = 0.37f;
    if (!(fabsf(y) > 1e-6f) || !(fabsf(y) < 1e4f)) y = -0.61f;
  }
  if (nbad) atomicAdd(&counts[0], nbad);
}

__global__ __launch_bounds__(256) void k_load(float* __restrict__ buf, int n, int iters) {   // plain fp32 traffic + ALU work: the "other process"
  const int i = blockIdx.x * 256 + threadIdx.x;
  float a = buf[i % n];
  for (int k = 0; k < iters; ++k) a = a * 1.0001f + buf[(i + k * 4099) % n] * 1e-3f;
  buf[i % n] = a;
}

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma(float* __restrict__ buf, int iters) {   // the matrix pipe kept busy, nothing else
  h16x4 a = {(h16)(threadIdx.x * 1e-3f), (h16)0.5f, (h16)-0.25f, (h16)0.125f}, b = {(h16)1.f, (h16)(blockIdx.x * 1e-4f), (h16)0.5f, (h16)-1.f};
  f32x16 c = {};
  for (int k = 0; k < iters; ++k) { c = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x8f16(b, a, c, 0, 0, 0); }
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += c[k];
  if (t == 12345.f) buf[0] = t;
}
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma16(float* __restrict__ buf, int iters) {   // gfx950's v_mfma_f32_32x32x16_f16
  h16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (h16)(threadIdx.x * 1e-3f + k); b[k] = (h16)(blockIdx.x * 1e-4f - k); }
  f32x16 c = {};
  for (int k = 0; k < iters; ++k) { c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0); }
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) t += c[k];
  if (t == 12345.f) buf[0] = t;
}
__global__ __launch_bounds__(256) void k_trload(float* __restrict__ buf, int iters) {   // ds_read_b64_tr_b16 in a loop
  __shared__ __attribute__((aligned(16))) short tile[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) tile[i] = (short)(i * 31);
  __syncthreads();
  int acc = 0;
  for (int k = 0; k < iters; ++k) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + ((threadIdx.x * 4 + k * 64) & 8188)));
    acc += v[0] + v[1] + v[2] + v[3];
  }
  if (acc == 12345) buf[0] = (float)acc;
}
__global__ __launch_bounds__(256) void k_gll(float* __restrict__ buf, int n, int iters) {   // global_load_lds_dwordx4 (direct to LDS) in a loop
  __shared__ __attribute__((aligned(16))) float tile[4096];
  const int wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (int k = 0; k < iters; ++k) {
    const float* src = buf + (((size_t)blockIdx.x * 256 + threadIdx.x) * 4 + (size_t)k * 65536) % (size_t)(n - 4);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(tile + wave * 256 + (k & 3) * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    acc += tile[(threadIdx.x + k) & 4095];
  }
  if (acc == 12345.f) buf[1] = acc;
}
__global__ __launch_bounds__(256) void k_valu(float* __restrict__ buf, int iters) {   // dependent fp32 FMAs from registers: VALU busy, no memory
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 0.5f;
  for (int k = 0; k < iters; ++k) { a = a * b + 0.25f; b = b * 0.999f + a * 1e-6f; }
  if (a == 12345.f) buf[0] = a + b;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
  const bool load = argc > 2 && (!strcmp(argv[2], "load") || !strcmp(argv[2], "mfma") || !strcmp(argv[2], "valu") || !strcmp(argv[2], "mfma16") ||
                                 !strcmp(argv[2], "trload") || !strcmp(argv[2], "gll"));
  const int src = argc > 2 && !load ? atoi(argv[2]) : 0;
  const bool same_process = argc > 3 && !strcmp(argv[3], "same");   // ./probe_pk_f32 <seconds> <src> same: the v_mfma_f32_32x32x16_f16 neighbour
                                                                    // runs on a second stream of THIS process instead of in another process
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  if (load) {
    const int n = 1 << 24; float* buf; (void)hipMalloc(&buf, n * sizeof(float)); (void)hipMemset(buf, 0, n * sizeof(float));
    long launches = 0;
    const int kind = !strcmp(argv[2], "mfma") ? 1 : !strcmp(argv[2], "valu") ? 2 : !strcmp(argv[2], "mfma16") ? 3 : !strcmp(argv[2], "trload") ? 4 : !strcmp(argv[2], "gll") ? 5 : 0;
    while (elapsed() < seconds) {
      for (int k = 0; k < 20; ++k) {
        if (kind == 1) hipLaunchKernelGGL(k_mfma, dim3(4096), dim3(256), 0, 0, buf, 2000);
        else if (kind == 3) hipLaunchKernelGGL(k_mfma16, dim3(4096), dim3(256), 0, 0, buf, 2000);
        else if (kind == 4) hipLaunchKernelGGL(k_trload, dim3(4096), dim3(256), 0, 0, buf, 4000);
        else if (kind == 5) hipLaunchKernelGGL(k_gll, dim3(4096), dim3(256), 0, 0, buf, n, 200);
        else if (kind == 2) hipLaunchKernelGGL(k_valu, dim3(8192), dim3(256), 0, 0, buf, 4000);
        else hipLaunchKernelGGL(k_load, dim3(8192), dim3(256), 0, 0, buf, n, 64);
      }
      (void)hipDeviceSynchronize(); launches += 20;
    }
    printf("%s role: %ld launches\n", argv[2], launches);
    return 0;
  }
  const int nblk = 1024, n = nblk * 512;
  float* h = (float*)malloc(n * sizeof(float));
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = ((rand() % 2001) - 1000) * 1e-3f * (1.0f / (1 << (rand() % 12)));
  float* xs; uint32_t* counts; Event* ev;
  const int NC = 128 + 4096;
  (void)hipMalloc(&xs, n * sizeof(float)); (void)hipMalloc(&counts, NC * 4); (void)hipMalloc(&ev, 64 * sizeof(Event));
  (void)hipMemcpy(xs, h, n * sizeof(float), hipMemcpyHostToDevice);
  (void)hipMemset(counts, 0, NC * 4);
  long launches = 0;
  hipStream_t sb = nullptr; float* nbuf = nullptr;
  if (same_process) { (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking); (void)hipMalloc(&nbuf, 1 << 20); }
  while (elapsed() < seconds) {
    if (same_process) for (int k = 0; k < 12; ++k) hipLaunchKernelGGL(k_mfma16, dim3(4096), dim3(256), 0, sb, nbuf, 2000);
    for (int k = 0; k < 50; ++k) {
      if (src == 0) hipLaunchKernelGGL(k_probe<0>, dim3(nblk), dim3(512), 0, 0, xs, counts, ev, 200);
      else if (src == 1) hipLaunchKernelGGL(k_probe<1>, dim3(nblk), dim3(512), 0, 0, xs, counts, ev, 200);
      else if (src == 2) hipLaunchKernelGGL(k_probe<2>, dim3(nblk), dim3(512), 0, 0, xs, counts, ev, 200);
      else hipLaunchKernelGGL((k_probe<1, true>), dim3(nblk), dim3(512), 0, 0, xs, counts, ev, 200);   // 3: the other packed families
    }
    (void)hipDeviceSynchronize();
    launches += 50;
  }
  uint32_t* hc = (uint32_t*)malloc(NC * 4); Event he[64];
  (void)hipMemcpy(hc, counts, NC * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(he, ev, sizeof(he), hipMemcpyDeviceToHost);
  if (same_process) printf("(neighbour: v_mfma_f32_32x32x16_f16 kernels on a second stream of this process)\n");
  printf("probe_pk_f32 (factor source %d): %ld launches x %d blocks x 512 lanes x 200 x 8 products: %u packed products differ from the scalar ones (%.3g of all)\n",
         src, launches, nblk, hc[0], hc[0] / (double(launches) * nblk * 512 * 200 * 8));
  if (hc[0]) {
    printf("  by lane:"); for (int i = 0; i < 64; ++i) if (hc[8 + i]) printf(" %d:%u", i, hc[8 + i]); printf("\n");
    if (src == 3) printf("  by form (pk_add_f32, pk_fma_f32, pk_mul_f16, pk_fma_f16; all with the crossed pick): %u %u %u %u\n", hc[72], hc[73], hc[74], hc[75]);
    else printf("  by form (w*g, w*g.yx, w.xx*g, w.yy*g.yx): %u %u %u %u\n", hc[72], hc[73], hc[74], hc[75]);
    int units = 0; for (int i = 0; i < 4096; ++i) units += hc[128 + i] != 0;
    printf("  SIMDs (xcc, se, cu, simd) with at least one wrong product: %d\n", units);
    int shown = 0;
    for (int i = 0; i < 4096 && shown < 24; ++i) if (hc[128 + i]) { printf("    xcc %d se %d cu %2d simd %d: %u\n", i >> 9, (i >> 6) & 7, (i >> 2) & 15, i & 3, hc[128 + i]); ++shown; }
    for (int i = 0; i < 12 && i < (int)hc[1]; ++i)
      printf("  event: lane %2u form %u  w %08x (%g) g %08x (%g)  packed %08x (%g)  scalar %08x (%g)  hw_id %08x xcc %u\n", he[i].lane, he[i].form, he[i].w,
             *(float*)&he[i].w, he[i].g, *(float*)&he[i].g, he[i].pk, *(float*)&he[i].pk, he[i].scalar, *(float*)&he[i].scalar, he[i].hwid, he[i].xcc & 15u);
  }
  return hc[0] ? 1 : 0;
}
