// LDS bank-conflict model of gfx950 for 8-byte (ds_read_b64 / ds_write_b64) and 16-byte accesses:
// one wave, every lane reads address base[lane] (float2 index) ITER times through a dependent chain;
// cycles per access for a set of lane -> address maps (unit stride, the padded maps of rdsp_fft.h,
// strides).  Build: hipcc --offload-arch=gfx950 -O3 lds_banks.hip -o lds_banks ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>
#include <string>

constexpr int ITER = 4096;

__global__ void __launch_bounds__(64) k_read64(const int *idx, long long *cycles, float *sink) {
  __shared__ float2 buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = make_float2((float)i, 0.f);
  __syncthreads();
  int a = idx[threadIdx.x];
  float acc = 0.f;
  long long t0 = clock64();
#pragma unroll 8
  for (int it = 0; it < ITER; it++) {
    float2 v = buf[a];
    acc += v.y;                     // v.y == 0: keeps the address chain dependent without changing it
    a += (int)v.y;
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = acc + (float)a;
}
__global__ void __launch_bounds__(64) k_write64(const int *idx, long long *cycles, float *sink) {
  __shared__ float2 buf[4096];
  int a = idx[threadIdx.x];
  long long t0 = clock64();
#pragma unroll 8
  for (int it = 0; it < ITER; it++) {
    buf[a] = make_float2((float)it, 1.f);
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
  }
  long long t1 = clock64();
  __syncthreads();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 64 + threadIdx.x] = buf[threadIdx.x].x;
}

int main() {
  int *d_idx; long long *d_c; float *d_s;
  hipMalloc(&d_idx, 64 * sizeof(int)); hipMalloc(&d_c, sizeof(long long)); hipMalloc(&d_s, 64 * sizeof(float));
  struct Map { std::string name; std::function<int(int)> f; };
  std::vector<Map> maps = {
    {"unit stride            t", [](int t) { return t; }},
    {"pad 1/4   t + t/4       ", [](int t) { return t + t / 4; }},
    {"pad 1/8   t + t/8       ", [](int t) { return t + t / 8; }},
    {"pad 1/16  t + t/16      ", [](int t) { return t + t / 16; }},
    {"pad 1/32  t + t/32      ", [](int t) { return t + t / 32; }},
    {"stride 2                ", [](int t) { return 2 * t; }},
    {"stride 4                ", [](int t) { return 4 * t; }},
    {"stride 5 (last pass P=4)", [](int t) { return 5 * t; }},
    {"stride 8                ", [](int t) { return 8 * t; }},
    {"stride 9 (last pass P=8)", [](int t) { return 9 * t; }},
    {"stride 16               ", [](int t) { return 16 * t; }},
    {"stride 17 (last, P=16)  ", [](int t) { return 17 * t; }},
    {"span 16 P=4 padded      ", [](int t) { int b = (t / 16) * 64 + t % 16; return b + b / 4; }},
    {"span 16 P=4 unpadded    ", [](int t) { return (t / 16) * 64 + t % 16; }},
    {"span 4 P=4 padded       ", [](int t) { int b = (t / 4) * 16 + t % 4; return b + b / 4; }},
    {"span 4 P=4 unpadded     ", [](int t) { return (t / 4) * 16 + t % 4; }},
    {"span 8 P=8 padded       ", [](int t) { int b = (t / 8) * 64 + t % 8; return b + b / 8; }},
    {"span 8 P=8 unpadded     ", [](int t) { return (t / 8) * 64 + t % 8; }},
    {"same address            ", [](int t) { return 7; }},
    {"two addresses 32 apart  ", [](int t) { return (t & 1) * 16; }},
    {"lanes 0-31 unit, 32-63 +16", [](int t) { return t < 32 ? t : t + 16; }},
  };
  for (auto &m : maps) {
    int h[64];
    for (int t = 0; t < 64; t++) h[t] = m.f(t);
    hipMemcpy(d_idx, h, sizeof(h), hipMemcpyHostToDevice);
    long long cr = 0, cw = 0;
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_read64, dim3(1), dim3(64), 0, 0, d_idx, d_c, d_s);
      hipMemcpy(&cr, d_c, sizeof(cr), hipMemcpyDeviceToHost);
      hipLaunchKernelGGL(k_write64, dim3(1), dim3(64), 0, 0, d_idx, d_c, d_s);
      hipMemcpy(&cw, d_c, sizeof(cw), hipMemcpyDeviceToHost);
    }
    printf("%s  read %.2f  write %.2f  (clock64 ticks per access)\n", m.name.c_str(), (double)cr / ITER, (double)cw / ITER);
  }
  return 0;
}
