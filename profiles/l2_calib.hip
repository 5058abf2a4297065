// l2_calib.hip — what one TCP_TCC_READ_REQ is worth in bytes on gfx950, and how many of them per second the L2s serve for the
// access shapes the likelihood kernel uses (`L2_LINE_BYTES = 128` was assumed, not calibrated).
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/l2_calib profiles/l2_calib.hip
//   gpurun_out/l2_calib                         # times every kernel with hipEvents (achieved request-side rates)
//   rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum ... -- gpurun_out/l2_calib 1
//
// Kernels (every one reads a KNOWN number of bytes; the sum of what was read goes to `sink` so nothing is elided):
//   stream16      coalesced: lane l of a wavefront reads 16 B at base + 16 l (1 KB per load instruction), 1 GiB in total, HBM
//   gather64_lane every lane reads the four 16-B parts of ITS OWN 64-byte record (four load instructions, 64 lines each)
//   gather64_quad the cooperative fetch of likelihood_kernels.h: lane j of a quad reads part j of the records of the
//                 quad's four lanes (four load instructions, 16 whole records each)
//   gather128_oct lane j of an octet reads part j of a 128-byte record (eight loads cover the octet's eight records)
// each gather over a 2 MB set (resident in every XCD's 4 MB L2, 64x the 32 KB L1) and over a 1 GiB set (HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ buf, size_t n_vec, float* sink)
{
  float acc = 0.f;
  for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < n_vec; i += size_t(gridDim.x) * 256)
  {
    const float4 v = buf[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *sink = acc;
}

// rounds x records per lane; record index = hash of (global lane, round) modulo n_rec
template <int SET>  // SET only separates the 2 MB and the 1 GiB launches by kernel name in the profiler output
__global__ void __launch_bounds__(256) gather64_lane(const float4* __restrict__ buf, uint32_t rec_mask, int rounds, float* sink)
{
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
  {
    const uint32_t rec = hash32(gid * 977u + r * 0x9e3779b9u) & rec_mask;
    const float4* p = buf + size_t(rec) * 4;
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x + b.y + c.z + d.w;
  }
  if (acc == 123.456f) *sink = acc;
}

template <int SET>  // SET only separates the 2 MB and the 1 GiB launches by kernel name in the profiler output
__global__ void __launch_bounds__(256) gather64_quad(const float4* __restrict__ buf, uint32_t rec_mask, int rounds, float* sink)
{
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t j = threadIdx.x & 3, q0 = gid & ~3u;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
  {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const uint32_t rec = hash32((q0 + k) * 977u + r * 0x9e3779b9u) & rec_mask;  // the record of lane k of this quad
      v[k] = buf[size_t(rec) * 4 + j];
    }
    acc += v[0].x + v[1].y + v[2].z + v[3].w;
  }
  if (acc == 123.456f) *sink = acc;
}

template <int SET>  // SET only separates the 2 MB and the 1 GiB launches by kernel name in the profiler output
__global__ void __launch_bounds__(256) gather128_oct(const float4* __restrict__ buf, uint32_t rec_mask, int rounds, float* sink)
{
  const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
  const uint32_t j = threadIdx.x & 7, o0 = gid & ~7u;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
  {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
    {
      const uint32_t rec = hash32((o0 + k) * 977u + r * 0x9e3779b9u) & rec_mask;
      v[k] = buf[size_t(rec) * 8 + j];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k].x;
  }
  if (acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv)
{
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const size_t big = size_t(1) << 30;
  float4* buf; float* sink;
  CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 0, big));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * 8, rounds = 64;
  const double lanes = double(grid) * 256;
  struct Case { const char* name; int kind; size_t set_bytes; };
  const Case cases[] = { { "stream16 1GiB", 0, big },
                         { "gather64_lane 2MB", 1, size_t(2) << 20 }, { "gather64_quad 2MB", 2, size_t(2) << 20 },
                         { "gather128_oct 2MB", 3, size_t(2) << 20 },
                         { "gather64_lane 1GiB", 1, big }, { "gather64_quad 1GiB", 2, big }, { "gather128_oct 1GiB", 3, big } };
  printf("%-22s %10s %14s %14s %12s\n", "kernel", "ms", "bytes read", "records", "GB/s");
  for (const Case& c : cases)
  {
    float best = 1e30f;
    for (int it = 0; it < reps + 1; ++it)
    {
      CK(hipEventRecord(e0));
      switch (c.kind)
      {
        case 0: hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, buf, big / 16, sink); break;
        case 1: if (c.set_bytes == big) hipLaunchKernelGGL(gather64_lane<1>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 64 - 1), rounds, sink); else hipLaunchKernelGGL(gather64_lane<0>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 64 - 1), rounds, sink); break;
        case 2: if (c.set_bytes == big) hipLaunchKernelGGL(gather64_quad<1>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 64 - 1), rounds, sink); else hipLaunchKernelGGL(gather64_quad<0>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 64 - 1), rounds, sink); break;
        case 3: if (c.set_bytes == big) hipLaunchKernelGGL(gather128_oct<1>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 128 - 1), rounds, sink); else hipLaunchKernelGGL(gather128_oct<0>, dim3(grid), dim3(256), 0, 0, buf, uint32_t(c.set_bytes / 128 - 1), rounds, sink); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) best = ms < best ? ms : best;
    }
    const double recs = c.kind == 0 ? 0 : lanes * rounds;
    const double bytes = c.kind == 0 ? double(big) : recs * (c.kind == 3 ? 128.0 : 64.0);
    printf("%-22s %10.4f %14.0f %14.0f %12.1f\n", c.name, best, bytes, recs, bytes / (best * 1e-3) / 1e9);
  }
  return 0;
}
