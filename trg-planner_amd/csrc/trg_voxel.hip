// trg_voxel.hip -- voxel-grid downsampling of the prebuilt map on the GPU (gfx950).
//
// Replaces the pcl::VoxelGrid step of TRGPlanner::loadPrebuiltMap (trg_planner.cpp:91-94): one
// output point per occupied leaf^3 voxel, the centroid of the points inside it, voxels emitted in
// ascending voxel index.  PCL is not vendored by the reference, so this follows PCL's published
// algorithm (filters/impl/voxel_grid.hpp, applyFilter): fp32 bounds, min_b = floor(min * inv_leaf),
// ijk = (int)(floor(p * inv_leaf) - (float)min_b), idx = i + j*dx + k*dx*dy, sort by idx, fp32 sum of
// each run divided by its count.  PCL's std::sort leaves the order inside a voxel unspecified; here
// it is ascending point index, so the fp32 sums are deterministic.
//
// HBM-bound integer/byte work: key build (read 12 B, write 8 B per point), one 64-bit radix sort
// (rocPRIM), head flags + scan, and a gather-sum per voxel.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "trg_kernels.h"

namespace trg {
namespace {

__device__ __forceinline__ unsigned ord_key(float f) {
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline float ord_float(unsigned k) {
  unsigned b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

// bounds[0..2] = min keys, bounds[3..5] = max keys over the finite points
__global__ __launch_bounds__(256) void k_vox_bounds(const float *xyz, size_t n, size_t stride,
                                                    unsigned *bounds) {
  unsigned mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) continue;  // pcl::getMinMax3D, !is_dense
    const unsigned k[3] = {ord_key(x), ord_key(y), ord_key(z)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = min(mn[a], k[a]);
      mx[a] = max(mx[a], k[a]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      mn[a] = min(mn[a], (unsigned)__shfl_xor((int)mn[a], m));
      mx[a] = max(mx[a], (unsigned)__shfl_xor((int)mx[a], m));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      atomicMin(&bounds[a], mn[a]);
      atomicMax(&bounds[3 + a], mx[a]);
    }
  }
}

struct VoxGrid {
  float inv;          // 1 / leaf
  int min_b[3];       // floor(min * inv)
  int mul1, mul2;     // dx, dx*dy
};

// key = voxel index << 32 | point index; non-finite points get the all-ones key (sorted last)
__global__ __launch_bounds__(256) void k_vox_keys(const float *xyz, size_t n, size_t stride,
                                                  VoxGrid g, unsigned long long *keys,
                                                  unsigned *bad_count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const float x = xyz[i * stride], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    unsigned long long k = ~0ull;
    if (isfinite(x) && isfinite(y) && isfinite(z)) {
      const int i0 = (int)(floorf(x * g.inv) - (float)g.min_b[0]);
      const int i1 = (int)(floorf(y * g.inv) - (float)g.min_b[1]);
      const int i2 = (int)(floorf(z * g.inv) - (float)g.min_b[2]);
      const int idx = i0 + i1 * g.mul1 + i2 * g.mul2;
      k = ((unsigned long long)(unsigned)idx << 32) | (unsigned long long)(unsigned)i;
    } else {
      atomicAdd(bad_count, 1u);
    }
    keys[i] = k;
  }
}

__global__ __launch_bounds__(256) void k_vox_heads(const unsigned long long *keys, size_t n,
                                                   int *head) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long k = keys[i];
    const bool valid = k != ~0ull;
    head[i] = (valid && (i == 0 || (keys[i - 1] >> 32) != (k >> 32))) ? 1 : 0;
  }
}

// seg[i] = exclusive scan of head; a head at sorted position i opens voxel seg[i]
__global__ __launch_bounds__(256) void k_vox_starts(const int *head, const int *seg, size_t n,
                                                    int *start) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    if (head[i]) start[seg[i]] = (int)i;
}

// one thread per voxel: fp32 sum in sorted (= ascending point index) order, then / (float)count
__global__ __launch_bounds__(256) void k_vox_centroid(const float *xyz, size_t stride,
                                                      const unsigned long long *keys,
                                                      const int *start, int nvox, int n_valid,
                                                      float *out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const int s = start[v], e = (v + 1 < nvox) ? start[v + 1] : n_valid;
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (int i = s; i < e; ++i) {
    const size_t p = (size_t)(keys[i] & 0xFFFFFFFFull);
    sx += xyz[p * stride];
    sy += xyz[p * stride + 1];
    sz += xyz[p * stride + 2];
  }
  const float c = (float)(e - s);
  out[3 * (size_t)v] = sx / c;
  out[3 * (size_t)v + 1] = sy / c;
  out[3 * (size_t)v + 2] = sz / c;
}

__global__ __launch_bounds__(256) void k_vox_copy(const float *xyz, size_t n, size_t stride,
                                                  float *out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    out[3 * i] = xyz[i * stride];
    out[3 * i + 1] = xyz[i * stride + 1];
    out[3 * i + 2] = xyz[i * stride + 2];
  }
}

inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

#define VOX_TRY(x)                    \
  do {                                \
    hipError_t e_ = (x);              \
    if (e_ != hipSuccess) {           \
      release();                      \
      return e_;                      \
    }                                 \
  } while (0)

}  // namespace

// d_xyz: n points (device), d_out: room for 3*n floats (device).  *status: 0 = filtered,
// 1 = leaf too small for 32-bit voxel indices -> input copied unchanged (PCL's behaviour).
hipError_t voxel_grid_filter(const float *d_xyz, size_t n, size_t stride, float leaf, float *d_out,
                             size_t *n_out, int *status, hipStream_t s) {
  *n_out = 0;
  *status = 0;
  if (n == 0) return hipSuccess;
  if (n >= 0x7FFFFFFFull) return hipErrorInvalidValue;
  unsigned *d_bounds = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  int *d_head = nullptr, *d_seg = nullptr, *d_start = nullptr, *d_tmp = nullptr;
  void *d_sort_tmp = nullptr;
  auto release = [&]() {
    void *ptrs[] = {d_bounds, d_keys, d_keys2, d_head, d_seg, d_start, d_tmp, d_sort_tmp};
    for (void *p : ptrs)
      if (p) (void)hipFree(p);
  };
  VOX_TRY(hipMalloc((void **)&d_bounds, 6 * sizeof(unsigned)));
  const unsigned init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  VOX_TRY(hipMemcpyAsync(d_bounds, init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_vox_bounds, dim3(grid_for(n)), dim3(256), 0, s, d_xyz, n, stride, d_bounds);
  unsigned hb[6];
  VOX_TRY(hipMemcpyAsync(hb, d_bounds, sizeof(hb), hipMemcpyDeviceToHost, s));
  VOX_TRY(hipStreamSynchronize(s));
  if (hb[0] == 0xFFFFFFFFu && hb[3] == 0u) {  // no finite point at all
    release();
    return hipSuccess;
  }
  float mn[3], mx[3];
  for (int a = 0; a < 3; ++a) {
    mn[a] = ord_float(hb[a]);
    mx[a] = ord_float(hb[3 + a]);
  }
  VoxGrid g;
  g.inv = 1.0f / leaf;
  // PCL: refuse when the index space overflows int32 and hand the input through
  const int64_t dx = (int64_t)((mx[0] - mn[0]) * g.inv) + 1;
  const int64_t dy = (int64_t)((mx[1] - mn[1]) * g.inv) + 1;
  const int64_t dz = (int64_t)((mx[2] - mn[2]) * g.inv) + 1;
  if (dx * dy * dz > (int64_t)INT32_MAX) {
    hipLaunchKernelGGL(k_vox_copy, dim3(grid_for(n)), dim3(256), 0, s, d_xyz, n, stride, d_out);
    VOX_TRY(hipStreamSynchronize(s));
    *n_out = n;
    *status = 1;
    release();
    return hipSuccess;
  }
  int max_b[3];
  for (int a = 0; a < 3; ++a) {
    g.min_b[a] = (int)floorf(mn[a] * g.inv);
    max_b[a] = (int)floorf(mx[a] * g.inv);
  }
  g.mul1 = max_b[0] - g.min_b[0] + 1;
  g.mul2 = g.mul1 * (max_b[1] - g.min_b[1] + 1);

  VOX_TRY(hipMalloc((void **)&d_keys, n * sizeof(unsigned long long)));
  VOX_TRY(hipMalloc((void **)&d_keys2, n * sizeof(unsigned long long)));
  VOX_TRY(hipMemsetAsync(d_bounds, 0, sizeof(unsigned), s));  // reused as the non-finite counter
  hipLaunchKernelGGL(k_vox_keys, dim3(grid_for(n)), dim3(256), 0, s, d_xyz, n, stride, g, d_keys,
                     d_bounds);
  size_t tmp_bytes = 0;
  VOX_TRY(rocprim::radix_sort_keys(nullptr, tmp_bytes, d_keys, d_keys2, n, 0, 64, s));
  VOX_TRY(hipMalloc(&d_sort_tmp, tmp_bytes ? tmp_bytes : 8));
  VOX_TRY(rocprim::radix_sort_keys(d_sort_tmp, tmp_bytes, d_keys, d_keys2, n, 0, 64, s));

  VOX_TRY(hipMalloc((void **)&d_head, (n + 1) * sizeof(int)));
  VOX_TRY(hipMalloc((void **)&d_seg, (n + 1) * sizeof(int)));
  VOX_TRY(hipMalloc((void **)&d_tmp, (n / 2048 + 8) * sizeof(int)));
  hipLaunchKernelGGL(k_vox_heads, dim3(grid_for(n)), dim3(256), 0, s, d_keys2, n, d_head);
  launch_exclusive_scan(d_head, d_seg, (int)n, d_tmp, s);  // d_seg[n] = number of voxels
  int nvox = 0;
  VOX_TRY(hipMemcpyAsync(&nvox, d_seg + n, sizeof(int), hipMemcpyDeviceToHost, s));
  VOX_TRY(hipStreamSynchronize(s));
  if (nvox > 0) {
    VOX_TRY(hipMalloc((void **)&d_start, ((size_t)nvox + 1) * sizeof(int)));
    hipLaunchKernelGGL(k_vox_starts, dim3(grid_for(n)), dim3(256), 0, s, d_head, d_seg, n, d_start);
    int bad = 0;  // non-finite points: their all-ones keys sort behind the last voxel
    VOX_TRY(hipMemcpyAsync(&bad, d_bounds, sizeof(int), hipMemcpyDeviceToHost, s));
    VOX_TRY(hipStreamSynchronize(s));
    const int n_valid = (int)n - bad;
    hipLaunchKernelGGL(k_vox_centroid, dim3((nvox + 255) / 256), dim3(256), 0, s, d_xyz, stride,
                       d_keys2, d_start, nvox, n_valid, d_out);
    VOX_TRY(hipStreamSynchronize(s));
    VOX_TRY(hipGetLastError());
  }
  *n_out = (size_t)nvox;
  release();
  return hipSuccess;
}

}  // namespace trg
