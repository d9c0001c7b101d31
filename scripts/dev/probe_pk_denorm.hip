// Are the packed fp32 instructions bit-identical to the scalar ones on denormal inputs / results and in rounding?  (Why the
// scalar-fp32 build's 1500-step quality runs end at other numbers than the packed build's: profiles/r05_pk_f32_probe.txt, section 5.)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o probe_pk_denorm probe_pk_denorm.hip ; ./probe_pk_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, const float* b, const float* c, uint32_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f2 x = {a[i], a[(i + 1) % n]}, y = {b[i], b[(i + 1) % n]}, z = {c[i], c[(i + 1) % n]}, pm, pa, pf;
  asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pm) : "v"(x), "v"(y));
  asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pa) : "v"(x), "v"(y));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pf) : "v"(x), "v"(y), "v"(z));
  float sm, sa, sf;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(sm) : "v"(x.x), "v"(y.x));
  asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(sa) : "v"(x.x), "v"(y.x));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(sf) : "v"(x.x), "v"(y.x), "v"(z.x));
  out[6 * i + 0] = __float_as_uint(pm.x); out[6 * i + 1] = __float_as_uint(sm);
  out[6 * i + 2] = __float_as_uint(pa.x); out[6 * i + 3] = __float_as_uint(sa);
  out[6 * i + 4] = __float_as_uint(pf.x); out[6 * i + 5] = __float_as_uint(sf);
}
int main() {
  const int n = 1 << 20;
  float *ha = new float[n], *hb = new float[n], *hc = new float[n];
  uint32_t s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return s; };
  for (int i = 0; i < n; ++i) {
    uint32_t ua = rnd(), ub = rnd(), uc = rnd();
    if (i % 4 == 0) { ua &= 0x807FFFFFu; }                       // denormal a
    if (i % 4 == 1) { ua = (ua & 0x807FFFFFu) | 0x20000000u; ub = (ub & 0x807FFFFFu) | 0x1F000000u; }   // product in the denormal range
    if (i % 4 == 2) { ua = (ua & 0x80FFFFFFu); ub = (ub & 0x80FFFFFFu); }    // tiny operands for the add
    ha[i] = *(float*)&ua; hb[i] = *(float*)&ub; hc[i] = *(float*)&uc;
    if (ha[i] != ha[i] || hb[i] != hb[i] || hc[i] != hc[i]) { ha[i] = 1.5f; hb[i] = -2.25f; hc[i] = 0.125f; }
  }
  float *a, *b, *c; uint32_t* o;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4); (void)hipMalloc(&c, n * 4); (void)hipMalloc(&o, n * 24);
  (void)hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, c, o, n);
  uint32_t* ho = new uint32_t[6 * n];
  (void)hipMemcpy(ho, o, n * 24, hipMemcpyDeviceToHost);
  long d[3] = {0, 0, 0}; int shown = 0;
  for (int i = 0; i < n; ++i)
    for (int k2 = 0; k2 < 3; ++k2)
      if (ho[6 * i + 2 * k2] != ho[6 * i + 2 * k2 + 1]) {
        ++d[k2];
        if (shown < 8) { ++shown; printf("  %s: a %08x b %08x c %08x  packed %08x scalar %08x\n", k2 == 0 ? "mul" : k2 == 1 ? "add" : "fma", *(uint32_t*)&ha[i], *(uint32_t*)&hb[i], *(uint32_t*)&hc[i], ho[6 * i + 2 * k2], ho[6 * i + 2 * k2 + 1]); }
      }
  printf("probe_pk_denorm: %d cases (random bit patterns, a quarter with denormal inputs, a quarter with denormal products): packed != scalar in mul %ld, add %ld, fma %ld\n", n, d[0], d[1], d[2]);
  return 0;
}
