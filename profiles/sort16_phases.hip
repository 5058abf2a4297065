// profiles/sort16_phases.hip — where the time of the one-launch 16-bit sort goes (mcl_3dl_amd/csrc/sort_kernels.h:
// rs_sort16_kernel): the same kernel body with wall-clock stamps (100 MHz) at its phase boundaries, thread 0 of every work-group.
// Build: hipcc --offload-arch=gfx950 -O3 -I mcl_3dl_amd/csrc -o sort16_phases.bin profiles/sort16_phases.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

#include "sort_kernels.h"

using namespace mcl3dl;

template <int VARIANT>   // 0: as shipped; 1: no LDS atomics in the counting loops (loads + keys only); 2: no global loads there
__global__ __launch_bounds__(RS_THREADS) void sort16_stamped(RsKeyGen kg, RsFinal fin, int n, uint32_t mask, long long* stamps)
{
  __shared__ __attribute__((aligned(16))) uint32_t bins[32768];
  __shared__ uint32_t wtot[RS_WAVES];
  constexpr int BATCH = 8;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const int own0 = static_cast<int>(blockIdx.x) * RS_THREADS;
  const uint32_t m16 = mask & 0xffffu;
  long long* st = stamps + 8 * blockIdx.x;
  if (t == 0) st[0] = wall_clock64();
  Rs16KeyCtx<RS_KEY_MORTON> kc;
  kc.init(kg);
  uint4* b4 = reinterpret_cast<uint4*>(bins);
  for (int k = t; k < 8192; k += RS_THREADS)
    b4[k] = make_uint4(0u, 0u, 0u, 0u);
  const int own = own0 + t;
  const bool have = own < n;
  uint32_t kraw = 0, val = 0;
  if (have)
  {
    kraw = kc.key(kg, static_cast<uint32_t>(own), kg.pts[own]);
    val = static_cast<uint32_t>(own);
  }
  const uint32_t k16 = kraw & m16;
  const uint32_t sh = (k16 & 1u) * 16u;
  const uint32_t own_word = rs16_word(k16 >> 1);
  uint32_t sink = 0;
  const auto count_range = [&](int first, int last)
  {
    for (int base = first + t; base < last; base += BATCH * RS_THREADS)
    {
      float4 p[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
      {
        const int idx = base + u * RS_THREADS;
        if (idx < last)
          p[u] = VARIANT == 2 ? make_float4(idx * 0.001f, idx * 0.002f, 1.f, 0.f) : kg.pts[idx];
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
      {
        const int idx = base + u * RS_THREADS;
        if (idx < last)
        {
          const uint32_t k = kc.key(kg, static_cast<uint32_t>(idx), p[u]) & m16;
          if (VARIANT == 1)
            sink += k;
          else
            atomicAdd(&bins[rs16_word(k >> 1)], 1u << ((k & 1u) * 16u));
        }
      }
    }
  };
  __syncthreads();
  if (t == 0) st[1] = wall_clock64();
  count_range(0, own0);
  const unsigned long long m = rs_match_bits<16>(k16, have);
  const uint32_t below = rs_lanes_below(m), total = static_cast<uint32_t>(__popcll(m));
  uint32_t tie = 0;
  __syncthreads();
  if (t == 0) st[2] = wall_clock64();
  for (int ww = 0; ww < RS_WAVES; ++ww)
  {
    if (w == ww)
    {
      uint32_t base = 0;
      if (have)
        base = (bins[own_word] >> sh) & 0xffffu;
      __builtin_amdgcn_wave_barrier();
      if (have && below + 1 == total)
        atomicAdd(&bins[own_word], total << sh);
      tie = base + below;
    }
    __syncthreads();
  }
  if (t == 0) st[3] = wall_clock64();
  count_range(min(own0 + RS_THREADS, n), n);
  __syncthreads();
  if (t == 0) st[4] = wall_clock64();
  {
    const uint32_t first = 2048u * static_cast<uint32_t>(w) + 32u * static_cast<uint32_t>(lane);
    uint32_t sum = 0;
#pragma unroll 8
    for (int r = 0; r < 32; ++r)
    {
      const uint32_t x = bins[rs16_word(first + r)];
      sum += (x & 0xffffu) + (x >> 16);
    }
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
    {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off)
        inc += o;
    }
    uint32_t run = inc - sum;
#pragma unroll 8
    for (int r = 0; r < 32; ++r)
    {
      const uint32_t x = bins[rs16_word(first + r)];
      const uint32_t lo = x & 0xffffu, hi = x >> 16;
      bins[rs16_word(first + r)] = run | ((run + lo) << 16);
      run += lo + hi;
    }
    if (lane == 63)
      wtot[w] = inc;
  }
  __syncthreads();
  if (t == 0) st[5] = wall_clock64();
  if (have)
  {
    uint32_t pos = ((bins[own_word] >> sh) & 0xffffu) + tie + (sink & 0u);
    for (uint32_t ww = 0; ww < (k16 >> 12); ++ww)
      pos += wtot[ww];
    if (VARIANT != 0)
      pos = static_cast<uint32_t>(own);
    float4 q = fin.src_pts[val];
    q.w = 0.f;
    fin.out_pts[pos] = q;
    fin.out_perm[pos] = val;
  }
  __syncthreads();
  if (t == 0) st[6] = wall_clock64();
}

__global__ void touch(float4* p, int n) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i].w = 0.f; }

int main()
{
  for (int n : { 16384, 4096, 32768 })
  {
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-15.f, 15.f), uz(-1.f, 3.f);
    std::vector<float4> h(n);
    for (auto& p : h) p = make_float4(u(rng), u(rng), uz(rng), 0.f);
    float mm[6] = { 1e9f, 1e9f, 1e9f, -1e9f, -1e9f, -1e9f };
    for (auto& p : h) { const float c[3] = { p.x, p.y, p.z }; for (int a = 0; a < 3; ++a) { mm[a] = std::min(mm[a], c[a]); mm[3 + a] = std::max(mm[3 + a], c[a]); } }
    float4 *d_in, *d_out; uint32_t* d_perm; float* d_mm; long long* d_st;
    hipMalloc(&d_in, sizeof(float4) * n); hipMalloc(&d_out, sizeof(float4) * n); hipMalloc(&d_perm, 4 * n); hipMalloc(&d_mm, 24);
    const int nb = (n + 1023) / 1024;
    hipMalloc(&d_st, 64 * nb);
    hipMemcpy(d_in, h.data(), sizeof(float4) * n, hipMemcpyHostToDevice); hipMemcpy(d_mm, mm, 24, hipMemcpyHostToDevice);
    RsKeyGen kg{}; kg.pts = d_in; kg.min3 = d_mm;
    RsFinal fin{ d_in, d_out, d_perm, 1 };
    const char* names[3] = { "as shipped", "no LDS atomics while counting", "no global loads while counting" };
    for (int variant = 0; variant < 3; ++variant)
    {
      std::vector<long long> st(8 * nb), acc(8, 0);
      const int reps = 20;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms_sum = 0;
      for (int rep = 0; rep < reps + 3; ++rep)
      {
        hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, n);   // the producer in front, as stage_pack is
        hipEventRecord(e0, 0);
        if (variant == 0) hipLaunchKernelGGL((sort16_stamped<0>), dim3(nb), dim3(1024), 0, 0, kg, fin, n, 0xffffu, d_st);
        if (variant == 1) hipLaunchKernelGGL((sort16_stamped<1>), dim3(nb), dim3(1024), 0, 0, kg, fin, n, 0xffffu, d_st);
        if (variant == 2) hipLaunchKernelGGL((sort16_stamped<2>), dim3(nb), dim3(1024), 0, 0, kg, fin, n, 0xffffu, d_st);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        if (rep < 3) continue;
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
        hipMemcpy(st.data(), d_st, 64 * nb, hipMemcpyDeviceToHost);
        // the slowest work-group of this launch, phase by phase
        int slow = 0;
        for (int b = 0; b < nb; ++b) if (st[8 * b + 6] - st[8 * b] > st[8 * slow + 6] - st[8 * slow]) slow = b;
        for (int k = 0; k < 6; ++k) acc[k] += st[8 * slow + k + 1] - st[8 * slow + k];
        long long first = st[0], last = st[6];
        for (int b = 0; b < nb; ++b) { first = std::min(first, st[8 * b]); last = std::max(last, st[8 * b + 6]); }
        acc[6] += last - first;
      }
      printf("n %5d (%2d work-groups) %-32s events %.2f us | first start -> last end %.2f us | slowest work-group: zero %.2f, count ahead %.2f, own (16 barriers) %.2f, "
             "count behind %.2f, prefix %.2f, place %.2f us\n", n, nb, names[variant], 1e3 * ms_sum / reps, acc[6] * 0.01 / reps, acc[0] * 0.01 / reps,
             acc[1] * 0.01 / reps, acc[2] * 0.01 / reps, acc[3] * 0.01 / reps, acc[4] * 0.01 / reps, acc[5] * 0.01 / reps);
    }
    hipFree(d_in); hipFree(d_out); hipFree(d_perm); hipFree(d_mm); hipFree(d_st);
  }
  return 0;
}
