// Micro-test: does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_f16_f32 saturate at +-65504 instead of producing inf on gfx950?
// (the split-fp16 converters of ds_device.h want a free saturating conversion).  Development tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

__global__ void k(const float* in, float* out, int n, int set) {
  if (set) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);   // hwreg(HW_REG_MODE, 23, 1) = 1
  const int i = threadIdx.x;
  if (i < n) {
    const float x = in[i];
    const _Float16 h1 = (_Float16)x;
    const _Float16 h2 = (_Float16)((x - (float)h1) * 2048.0f);
    out[2 * i] = (float)h1;
    out[2 * i + 1] = (float)h2;
  }
}

int main() {
  const int n = 10;
  float h[n] = {1.0f, 65504.0f, 65519.0f, 65520.0f, 7e4f, 1e6f, -1e6f, INFINITY, NAN, 3e38f};
  float *d, *o, r[2 * n];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int set = 0; set < 2; ++set) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n, set);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("ovfl=%d x=%g -> h1=%g h2=%g  recon=%g\n", set, h[i], r[2 * i], r[2 * i + 1], r[2 * i] + r[2 * i + 1] / 2048.0f);
  }
  return 0;
}
