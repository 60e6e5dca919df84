// Probe (GPU box): device-to-pinned-host rates of one copy, three concurrent copies and a copy kernel.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const float4 *src, float4 *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 81 * 1000 * 1000 / 48 * 48;
  char *d = nullptr, *h = nullptr;
  CK(hipMalloc((void **)&d, bytes));
  CK(hipMemset(d, 1, bytes));
  CK(hipHostMalloc((void **)&h, bytes, hipHostMallocDefault));
  memset(h, 0, bytes);
  hipStream_t s[3];
  for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  void *hd = nullptr;
  CK(hipHostGetDevicePointer(&hd, h, 0));
  for (int rep = 0; rep < 4; ++rep) {
    double t0 = now();
    CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s[0]));
    CK(hipStreamSynchronize(s[0]));
    double t1 = now();
    for (int k = 0; k < 3; ++k) CK(hipMemcpyAsync(h + k * (bytes / 3), d + k * (bytes / 3), bytes / 3, hipMemcpyDeviceToHost, s[k]));
    for (int k = 0; k < 3; ++k) CK(hipStreamSynchronize(s[k]));
    double t2 = now();
    hipLaunchKernelGGL(k_copy, dim3(512), dim3(256), 0, s[0], (const float4 *)d, (float4 *)hd, bytes / 16);
    CK(hipStreamSynchronize(s[0]));
    double t3 = now();
    hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, s[0], (const float4 *)d, (float4 *)hd, bytes / 16);
    CK(hipStreamSynchronize(s[0]));
    double t4 = now();
    printf("%zu MB: one copy %.2f ms (%.1f GB/s) | 3 streams %.2f ms (%.1f GB/s) | kernel 512 wg %.2f ms (%.1f GB/s) | kernel 64 wg %.2f ms (%.1f GB/s)\n",
           bytes / 1000000, t1 - t0, bytes / 1e6 / (t1 - t0), t2 - t1, bytes / 1e6 / (t2 - t1), t3 - t2, bytes / 1e6 / (t3 - t2), t4 - t3, bytes / 1e6 / (t4 - t3));
  }
  return 0;
}
