// float_chain.h — the reference's float recurrences, `score_like += dist * match_weight` over the scan
// (src/lidar_measurement_model_likelihood.cpp:120-134) and `sum += p.probability_` over the particles
// (include/mcl_3dl/pf.h:255-260), run by a whole wavefront instead of one lane, with the reference's bits.
//
// s_{i+1} = fl(s_i + t_i) is a dependent chain: ~5 cycles a term for one lane, 63 lanes idle. But while the running sum stays
// inside one binade [2^e, 2^(e+1)) — ulp u = 2^(e-23) — it is a multiple of u, and adding a term t < 2^(e-1) that is not a
// rounding tie gives  fl(s + t) = s + RN_u(t)  EXACTLY, where RN_u(t) = (M + t) - M with M = 1.5 * 2^e is t rounded to the
// nearest multiple of u: the rounding no longer depends on s. Sums of such multiples of u below 2^(e+1) are exact in any
// association, so 64 lanes take eight consecutive terms each, a DPP prefix sum places every lane's partial behind its
// predecessors', and ONE pass retires 512 terms. What the shortcut cannot decide is left to real float adds, in order:
//   * a rounding tie (|t - RN_u(t)| == u / 2: round-to-even looks at the parity of s / u),
//   * a term that is not small against s (t >= 2^(e-1)), negative, NaN or infinite,
//   * the step on which the sum leaves its binade (s + prefix >= 2^(e+1): the ulp changes)
// — the first lane with any of these is found by a ballot, everything in front of it is final, its eight terms are added
// serially (uniform LDS reads), and the next pass starts behind them with the new exponent. A sum of n positive terms
// crosses ~log2(n) binades and meets a handful of ties, so a chain of n terms costs ~n / 512 + log2(n) passes of ~80
// wave-instructions instead of n dependent adds; the result is the serial loop's float, bit for bit, for ANY input
// (whenever an assumption does not hold the code is the serial loop).  tests/cpp/float_chain_emul.cpp replays the algorithm
// on the CPU against the plain loop (ties, equal terms, denormals, huge terms, sign changes); tests/test_gpu_fuzz.py and
// test_gpu_parity.py compare the kernels that use it with the reference itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace mcl3dl
{
// inclusive prefix sum over the 64 lanes of a wavefront (DPP: shifts inside the rows of 16, then the row totals)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ inline float dpp_or_zero(float x)
{
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ inline float wave_scan_add(float v)
{
  v = v + dpp_or_zero<0x111, 0xf, true>(v);  // row_shr:1 (a lane without a source reads 0)
  v = v + dpp_or_zero<0x112, 0xf, true>(v);  // row_shr:2
  v = v + dpp_or_zero<0x114, 0xf, true>(v);  // row_shr:4
  v = v + dpp_or_zero<0x118, 0xf, true>(v);  // row_shr:8
  // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 (rows outside the mask keep `old` = 0)
  v = v + dpp_or_zero<0x142, 0xa, false>(v);
  v = v + dpp_or_zero<0x143, 0xc, false>(v);
  return v;
}

__device__ inline float chain_quad(float s, const float4 t)
{
  s = s + t.x;
  s = s + t.y;
  s = s + t.z;
  s = s + t.w;
  return s;
}

// k quads from row4[q] on, serially — eight LDS reads in flight ahead of the adds that consume them (a read per quad in the
// loop costs its ~120 cycles of latency every four adds: 40 cycles a term instead of 8)
__device__ inline float chain_serial(const float4* row4, int q, int k, float s)
{
  int j = 0;
  for (; j + 8 <= k; j += 8)
  {
    float4 t[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      t[c] = row4[q + j + c];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      s = chain_quad(s, t[c]);
  }
  if (j + 4 <= k)
  {
    const float4 a = row4[q + j], b = row4[q + j + 1], c = row4[q + j + 2], d = row4[q + j + 3];
    s = chain_quad(chain_quad(chain_quad(chain_quad(s, a), b), c), d);
    j += 4;
  }
  for (; j < k; ++j)
    s = chain_quad(s, row4[q + j]);
  return s;
}

// four terms against the binade constants: their roundings to the ulp (their sum into `a`), and whether all four are
// decidable without looking at s — in [0, 2^(e-1)) and not a rounding tie. Straight-line code: one wavefront alone issues an
// instruction every ~8 cycles whatever it depends on (profiles/r06c_chain_bench.txt), so instructions are what a pass costs.
// The range test is ONE unsigned compare of the largest bit pattern: a float in [+0, 2^(e-1)) has bits below those of
// 2^(e-1); anything negative (sign bit; -0 included: it then simply takes the serial path), infinite or NaN has more.
__device__ inline bool chain_classify(const float4 t, float M, float small, float hu, float& a)
{
  const float r0 = (M + t.x) - M, r1 = (M + t.y) - M, r2 = (M + t.z) - M, r3 = (M + t.w) - M;
  // |t - r| <= u / 2 for a term in range (and t - r is exact): a tie is the maximum reaching u / 2
  const float d = fmaxf(fmaxf(fabsf(t.x - r0), fabsf(t.y - r1)), fmaxf(fabsf(t.z - r2), fabsf(t.w - r3)));
  const uint32_t mx = max(max(__float_as_uint(t.x), __float_as_uint(t.y)), max(__float_as_uint(t.z), __float_as_uint(t.w)));
  a = ((r0 + r1) + r2) + r3;
  return (mx < __float_as_uint(small)) & (d < hu);
}

// s0 + row[0] + row[1] + ... + row[n - 1] as the float recurrence, by ONE wavefront (all 64 lanes converged; every lane
// returns the sum). row: 16-byte aligned, LDS or global; the floats row[n .. 4 * ceil(n / 4) + 4) must be readable and hold +0
// (chain_row_floats(n) floats in all: the last quad's padding and one quad of zeros behind it).
// The first CHAIN_HEAD4 quads are added serially: a sum of similar terms doubles at terms 2, 4, 8, ... — five more binades
// between term 8 and term 256 — and a pass that ends at a binade's edge after a few dozen terms costs more than those terms
// (a pass is ~90 instructions, i.e. ~90 serial adds: profiles/r06c_chain_bench.txt).
constexpr int CHAIN_HEAD4 = 64;
__host__ __device__ inline int chain_row_floats(int n)
{
  return ((n + 3) & ~3) + 4;
}
__device__ inline float seq_sum_wave(const float* row, int n, int lane, float s0 = 0.0f)
{
  const float4* row4 = reinterpret_cast<const float4*>(row);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n4 = (n + 3) >> 2;
  const int head = n4 < CHAIN_HEAD4 ? n4 : CHAIN_HEAD4;
  float s = chain_serial(row4, 0, head, s0);
  int q = head;
  if (q >= n4)
    return s;
  // this lane's eight consecutive terms of the pass that starts at quad q, read one pass ahead
  int mine = q + 2 * lane;
  float4 ta = mine < n4 ? row4[mine] : zero4, tb = mine + 1 < n4 ? row4[mine + 1] : zero4;
  while (q < n4)
  {
    s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));  // (uniform anyway: keeps the constants scalar)
    const int sb = __float_as_int(s) & 0x7f800000;
    int next;
    // a positive normal sum whose ulp / 2 and 2^(e+1) are normal floats; anything else (0, tiny, negative, inf, NaN) adds serially
    if (__float_as_int(s) < (40 << 23) || sb >= (254 << 23))
    {
      s = chain_quad(s, row4[q]);
      next = q + 1;
    }
    else
    {
      const float M = __int_as_float(sb | 0x00400000);     // 1.5 * 2^e
      const float small = __int_as_float(sb - (1 << 23));  // 2^(e-1)
      const float hu = __int_as_float(sb - (24 << 23));    // u / 2
      const float top = __int_as_float(sb + (1 << 23));    // 2^(e+1)
      // (the next pass's terms, in flight while this one computes: they are what it reads if nothing goes wrong here)
      const int ahead = mine + 128;
      const float4 na = ahead < n4 ? row4[ahead] : zero4, nb = ahead + 1 < n4 ? row4[ahead + 1] : zero4;
      float a0, a1;
      const bool ok = chain_classify(ta, M, small, hu, a0) & chain_classify(tb, M, small, hu, a1);
      const float p = wave_scan_add(a0 + a1);
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(!(ok & (s + p < top)));
      if (mask == 0ull)
      {
        s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), 63));
        q += 128;
        mine = ahead;
        ta = na;
        tb = nb;
        continue;
      }
      // everything in front of the first lane in trouble is final; its eight terms are added serially
      const int L = __builtin_ctzll(mask);
      if (L > 0)
        s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p), L - 1));
      int at = q + 2 * L;
      at = at < n4 ? at : n4 - 1;  // (a lane behind the row holds zeros and cannot be the first in trouble)
      const float4 x = row4[at], y = row4[at + 1];  // (at + 1 <= n4: the quad of zeros behind the row; x + 0.0f == x)
      s = chain_quad(chain_quad(s, x), y);
      next = at + 2;
    }
    q = next;
    mine = q + 2 * lane;
    ta = mine < n4 ? row4[mine] : zero4;
    tb = mine + 1 < n4 ? row4[mine + 1] : zero4;
  }
  return s;
}
}  // namespace mcl3dl
