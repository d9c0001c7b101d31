// Dev probe: what bounds the fused Adam stream (14.3 M parameters: read g, p, m, v; write g = 0, p, m, v, fp16 shadow)?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off probe_adam.hip -o probe_adam && ./probe_adam
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 h16;
typedef h16 h16x4 __attribute__((ext_vector_type(4)));
struct C { float inv_scale, step, isb, b1, b2, eps; };
__device__ inline void upd(float& g, float& p, float& m, float& v, h16& t, const C& c) {
  const float gi = g * c.inv_scale;
  const float mi = c.b1 * m + (1.f - c.b1) * gi, vi = c.b2 * v + (1.f - c.b2) * gi * gi;
  m = mi; v = vi;
  p -= c.step * (mi / (sqrtf(vi) * c.isb + c.eps));
  t = (h16)p;
}
__global__ void k_fill(float* g, size_t n, float s) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] = s * (float)((i * 2654435761u) >> 20); }
// V0: scalar grid-stride
__global__ void v0(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, h16* __restrict__ t, size_t n, C c) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i]; g[i] = 0.f; float pi = p[i], mi = m[i], vi = v[i]; h16 ti; upd(gi, pi, mi, vi, ti, c); m[i] = mi; v[i] = vi; p[i] = pi; t[i] = ti;
  }
}
// V2: vec4 grid-stride
__global__ void v2(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, h16* __restrict__ t, size_t n, C c) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n / 4; q += (size_t)gridDim.x * blockDim.x) {
    float4 g4 = ((float4*)g)[q], p4 = ((float4*)p)[q], m4 = ((float4*)m)[q], v4 = ((float4*)v)[q];
    ((float4*)g)[q] = make_float4(0, 0, 0, 0);
    h16x4 t4; h16 tt;
    upd(g4.x, p4.x, m4.x, v4.x, tt, c); t4[0] = tt; upd(g4.y, p4.y, m4.y, v4.y, tt, c); t4[1] = tt;
    upd(g4.z, p4.z, m4.z, v4.z, tt, c); t4[2] = tt; upd(g4.w, p4.w, m4.w, v4.w, tt, c); t4[3] = tt;
    ((float4*)m)[q] = m4; ((float4*)v)[q] = v4; ((float4*)p)[q] = p4; *(h16x4*)(t + 4 * q) = t4;
  }
}
// V3: vec4, block-contiguous: each block owns a contiguous span (consecutive iterations of a block touch consecutive memory)
__global__ void v3(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, h16* __restrict__ t, size_t n, C c) {
  const size_t nq = n / 4, per = (nq + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < nq ? lo + per : nq;
  for (size_t q = lo + threadIdx.x; q < hi; q += blockDim.x) {
    float4 g4 = ((float4*)g)[q], p4 = ((float4*)p)[q], m4 = ((float4*)m)[q], v4 = ((float4*)v)[q];
    ((float4*)g)[q] = make_float4(0, 0, 0, 0);
    h16x4 t4; h16 tt;
    upd(g4.x, p4.x, m4.x, v4.x, tt, c); t4[0] = tt; upd(g4.y, p4.y, m4.y, v4.y, tt, c); t4[1] = tt;
    upd(g4.z, p4.z, m4.z, v4.z, tt, c); t4[2] = tt; upd(g4.w, p4.w, m4.w, v4.w, tt, c); t4[3] = tt;
    ((float4*)m)[q] = m4; ((float4*)v)[q] = v4; ((float4*)p)[q] = p4; *(h16x4*)(t + 4 * q) = t4;
  }
}
// V4: vec4 with nontemporal accesses
__global__ void v4k(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, h16* __restrict__ t, size_t n, C c) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n / 4; q += (size_t)gridDim.x * blockDim.x) {
    float4 g4, p4, m4, v4;
    g4.x = __builtin_nontemporal_load(&g[4*q]); g4.y = __builtin_nontemporal_load(&g[4*q+1]); g4.z = __builtin_nontemporal_load(&g[4*q+2]); g4.w = __builtin_nontemporal_load(&g[4*q+3]);
    p4 = ((float4*)p)[q]; m4 = ((float4*)m)[q]; v4 = ((float4*)v)[q];
    ((float4*)g)[q] = make_float4(0, 0, 0, 0);
    h16x4 t4; h16 tt;
    upd(g4.x, p4.x, m4.x, v4.x, tt, c); t4[0] = tt; upd(g4.y, p4.y, m4.y, v4.y, tt, c); t4[1] = tt;
    upd(g4.z, p4.z, m4.z, v4.z, tt, c); t4[2] = tt; upd(g4.w, p4.w, m4.w, v4.w, tt, c); t4[3] = tt;
    ((float4*)m)[q] = m4; ((float4*)v)[q] = v4; ((float4*)p)[q] = p4; *(h16x4*)(t + 4 * q) = t4;
  }
}
// V5: plain copy-like upper bound: read 16 B, write 18 B per element with no math
__global__ void v5(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, h16* __restrict__ t, size_t n, C c) {
  for (size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x; q < n / 4; q += (size_t)gridDim.x * blockDim.x) {
    float4 g4 = ((float4*)g)[q], p4 = ((float4*)p)[q], m4 = ((float4*)m)[q], v4 = ((float4*)v)[q];
    ((float4*)g)[q] = make_float4(0, 0, 0, 0);
    p4.x += g4.x; m4.x += g4.y; v4.x += g4.z;
    h16x4 t4 = {(h16)p4.x, (h16)p4.y, (h16)p4.z, (h16)p4.w};
    ((float4*)m)[q] = m4; ((float4*)v)[q] = v4; ((float4*)p)[q] = p4; *(h16x4*)(t + 4 * q) = t4;
  }
}
int main() {
  const size_t n = 14292480;  // grid 14229504 + MLPs, multiple of 4
  float *p, *g, *m, *v; h16* t;
  hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&t, n * 2);
  hipMemset(p, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
  C c{1.f / 1024.f, 5e-3f, 1.f, 0.9f, 0.99f, 1e-15f};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int blocks, int threads, bool fill) {
    std::vector<float> ts;
    for (int r = 0; r < 25; ++r) {
      if (fill) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, g, n, 1e-6f);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, p, g, m, v, t, n, c);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 5) ts.push_back(ms * 1000.f);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-44s blocks %6d x %4d  fill %d : median %7.1f us  min %7.1f  (%.2f TB/s)\n", name, blocks, threads, (int)fill, ts[ts.size() / 2], ts[0], n * 34.0 / ts[ts.size() / 2] * 1e-6);
  };
  for (int fill = 0; fill < 2; ++fill) {
    run("v0 scalar grid-stride", v0, 4096, 256, fill);
    run("v0 scalar grid-stride", v0, 2048, 256, fill);
    run("v0 scalar one element per thread", v0, (int)((n + 255) / 256), 256, fill);
    run("v2 vec4 grid-stride", v2, 2048, 256, fill);
    run("v2 vec4 grid-stride", v2, 1024, 256, fill);
    run("v2 vec4 grid-stride", v2, 4096, 256, fill);
    run("v2 vec4 one group per thread", v2, (int)((n / 4 + 255) / 256), 256, fill);
    run("v3 vec4 block-contiguous", v3, 2048, 256, fill);
    run("v3 vec4 block-contiguous", v3, 1024, 256, fill);
    run("v3 vec4 block-contiguous 512 thr", v3, 1024, 512, fill);
    run("v4 vec4 nontemporal g", v4k, 2048, 256, fill);
    run("v5 copy-like bound (no math)", v5, 2048, 256, fill);
    run("v5 copy-like bound one group per thread", v5, (int)((n / 4 + 255) / 256), 256, fill);
  }
  return 0;
}
