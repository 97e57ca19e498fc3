#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline float rcp_refined(float d) { float r = __builtin_amdgcn_rcpf(d); float e = fmaf(-d, r, 1.f); return fmaf(e, r, r); }
__device__ inline float div_r(float n, float d, float r) { float q = n * r; float e = fmaf(-d, q, n); q = fmaf(e, r, q); e = fmaf(-d, q, n); return fmaf(e, r, q); }
__device__ inline uint32_t rng(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__global__ void k(unsigned long long* bad, int mode, float* ex) {
  uint32_t s = 1234567u + blockIdx.x * 9781u + threadIdx.x * 7919u + mode * 31u;
  unsigned long long nb = 0;
  for (int i = 0; i < 4096; ++i) {
    // numerator: log-uniform over ~[1e-20, 1e6]; divisor: integer counts 1..2^20 (mode 0), arbitrary positive floats (mode 1)
    const float mant = 1.f + (rng(s) >> 9) * (1.f / 8388608.f);
    const int ex2 = mode == 2 ? -(int)(rng(s) % 110u) - 40 : (int)(rng(s) % 86u) - 66;
    const float n = ldexpf(mant, ex2);
    float d;
    if (mode == 0 || mode == 2) d = (float)(1 + (rng(s) & 0xFFFFF));
    else d = ldexpf(1.f + (rng(s) >> 9) * (1.f / 8388608.f), (int)(rng(s) % 40u) - 10);
    const float ref = n / d;
    const float got = div_r(n, d, rcp_refined(d));
    if (__float_as_uint(ref) != __float_as_uint(got)) { if (nb == 0) { ex[0] = n; ex[1] = d; ex[2] = ref; ex[3] = got; } ++nb; }
  }
  atomicAdd(bad, nb);
}
int main() {
  unsigned long long* bad; float* ex;
  hipMalloc(&bad, 8); hipMalloc(&ex, 16);
  for (int mode = 0; mode < 3; ++mode) {
    hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(k, dim3(65536), dim3(256), 0, 0, bad, mode, ex);
    unsigned long long h; float he[4];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 16, hipMemcpyDeviceToHost);
    printf("mode %d: %llu mismatches of %llu  (example n=%.9g d=%.9g ref=%.9g got=%.9g)\n", mode, h, 65536ull * 256 * 4096, he[0], he[1], he[2], he[3]);
  }
  return 0;
}
