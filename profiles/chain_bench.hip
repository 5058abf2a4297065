// profiles/chain_bench.hip — what the reference's float recurrence over n terms in LDS costs one wavefront: the plain
// dependent chain (one lane, sixteen terms read ahead) against float_chain.h:seq_sum_wave (256 terms a pass), in shader
// cycles (s_memtime) inside the kernel, one work-group alone on its CU and 2048 of them at once.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../mcl_3dl_amd/csrc -o chain_bench.bin chain_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <random>
#include <vector>

#include "float_chain.h"

using namespace mcl3dl;

__device__ inline float serial_chain(const float* row, int n)
{
  float s = 0.0f;
  const float4* r4 = reinterpret_cast<const float4*>(row);
  const int n4 = (n + 3) >> 2;
  int q = 0;
  for (; q + 4 <= n4; q += 4)
  {
    const float4 a = r4[q], b = r4[q + 1], c = r4[q + 2], d = r4[q + 3];
    s = chain_quad(chain_quad(chain_quad(chain_quad(s, a), b), c), d);
  }
  for (; q < n4; ++q)
    s = chain_quad(s, r4[q]);
  return s;
}

template <int MODE>
__global__ __launch_bounds__(64) void bench_kernel(const float* terms, int n, float* out, long long* cycles)
{
  extern __shared__ __attribute__((aligned(16))) float row[];
  const int n4 = chain_row_floats(n);
  for (int i = threadIdx.x; i < n4; i += 64)
    row[i] = i < n ? terms[i] : 0.0f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  float s;
  if (MODE == 0)
    s = serial_chain(row, n);
  else if (MODE == 2)
  {
    // the plain chain with ONE active lane (does a wave64 instruction whose upper half is inactive issue faster?)
    s = 0.0f;
    if (threadIdx.x == 0)
      s = serial_chain(row, n);
    s = __shfl(s, 0);
  }
  else if (MODE == 3)
  {
    s = 0.0f;
    if (threadIdx.x < 32)
      s = serial_chain(row, n);
    s = __shfl(s, 0);
  }
  else
    s = seq_sum_wave(row, n, threadIdx.x);
  const long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0)
  {
    out[blockIdx.x] = s;
    cycles[blockIdx.x] = t1 - t0;
  }
}

int main()
{
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  float* d_terms;
  float* d_out;
  long long* d_cyc;
  hipMalloc(&d_terms, 70000 * 4);
  hipMalloc(&d_out, 4096 * 4);
  hipMalloc(&d_cyc, 4096 * 8);
  for (int n : { 96, 256, 512, 1000, 2048, 4096, 12288 })
  {
    std::vector<float> t(n);
    for (int i = 0; i < n; ++i)
    {
      const float d = 0.25f * U(rng);
      const float dist = 0.2f - (d > 0.05f ? d : 0.05f);
      t[i] = (dist < 0.f || (rng() & 3) == 0) ? 0.f : dist * 5.f;
    }
    hipMemcpy(d_terms, t.data(), n * 4, hipMemcpyHostToDevice);
    for (int blocks : { 1, 2048 })
    {
      double cyc[2];
      float res[2];
      float ms[2];
      for (int mode = 0; mode < 2; ++mode)
      {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const size_t lds = chain_row_floats(n) * 4;
        for (int rep = 0; rep < 3; ++rep)
        {
          hipEventRecord(e0);
          if (mode == 0)
            hipLaunchKernelGGL(bench_kernel<0>, dim3(blocks), dim3(64), lds, 0, d_terms, n, d_out, d_cyc);
          else
            hipLaunchKernelGGL(bench_kernel<1>, dim3(blocks), dim3(64), lds, 0, d_terms, n, d_out, d_cyc);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms[mode], e0, e1);
        std::vector<long long> c(blocks);
        hipMemcpy(c.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&res[mode], d_out, 4, hipMemcpyDeviceToHost);
        double sum = 0;
        for (long long v : c)
          sum += v;
        cyc[mode] = sum / blocks;
      }
      printf("n %5d, %4d work-groups: serial %8.0f cycles (%.2f / term), launch %.1f us | wavefront %8.0f cycles (%.2f / term), launch %.1f us | x %.2f | same bits %d\n",
             n, blocks, cyc[0], cyc[0] / n, ms[0] * 1e3, cyc[1], cyc[1] / n, ms[1] * 1e3, cyc[0] / cyc[1], res[0] == res[1]);
    }
  }
  // one active lane / the lower half only
  for (int n : { 256, 1000 })
  {
    std::vector<float> t(n, 0.4f);
    hipMemcpy(d_terms, t.data(), n * 4, hipMemcpyHostToDevice);
    const size_t lds = chain_row_floats(n) * 4;
    long long c0, c2, c3;
    hipLaunchKernelGGL(bench_kernel<0>, dim3(1), dim3(64), lds, 0, d_terms, n, d_out, d_cyc);
    hipMemcpy(&c0, d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(bench_kernel<2>, dim3(1), dim3(64), lds, 0, d_terms, n, d_out, d_cyc);
    hipMemcpy(&c2, d_cyc, 8, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(bench_kernel<3>, dim3(1), dim3(64), lds, 0, d_terms, n, d_out, d_cyc);
    hipMemcpy(&c3, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("n %5d serial chain: all 64 lanes %lld cycles | lane 0 only %lld | lanes 0-31 %lld\n", n, c0, c2, c3);
  }
  return 0;
}
