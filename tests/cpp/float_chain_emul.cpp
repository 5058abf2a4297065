// float_chain_emul.cpp — CPU replay of mcl_3dl_amd/csrc/float_chain.h:seq_sum_wave (the 64 lanes as a loop, the DPP prefix sum
// as a plain inclusive scan in the same association) against the plain float recurrence it must reproduce bit for bit
// (score_like += dist * match_weight, src/lidar_measurement_model_likelihood.cpp:120-134). Test infrastructure only:
// tests/test_float_chain_cpu.py builds and runs it (g++ -O2 -ffp-contract=off; x86-64 SSE floats round to nearest even like
// the device). Exit code 0 = every case equal; prints the pass statistics.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

static inline int f2i(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float i2f(int i) { float f; memcpy(&f, &i, 4); return f; }

static float serial(const std::vector<float>& t, float s0)
{
  volatile float s = s0;
  for (float x : t)
    s = s + x;
  return s;
}

static long g_passes = 0, g_serial_quads = 0, g_clean_passes = 0;

// the inclusive scan in the association of wave_scan_add: Hillis-Steele steps 1, 2, 4, 8 inside rows of 16, then the row totals
static void scan64(float* v)
{
  for (int d = 1; d <= 8; d <<= 1)
  {
    float o[64];
    for (int l = 0; l < 64; ++l)
      o[l] = (l % 16) >= d ? v[l - d] : 0.0f;
    for (int l = 0; l < 64; ++l)
      v[l] = v[l] + o[l];
  }
  {
    float o[64];
    for (int l = 0; l < 64; ++l)
      o[l] = (l / 16 == 1) ? v[15] : (l / 16 == 3) ? v[47] : 0.0f;
    for (int l = 0; l < 64; ++l)
      v[l] = v[l] + o[l];
  }
  {
    float o[64];
    for (int l = 0; l < 64; ++l)
      o[l] = (l / 16 >= 2) ? v[31] : 0.0f;
    for (int l = 0; l < 64; ++l)
      v[l] = v[l] + o[l];
  }
}

static float quad(float s, const float* t)
{
  volatile float v = s;
  v = v + t[0];
  v = v + t[1];
  v = v + t[2];
  v = v + t[3];
  return v;
}

// chain_classify: four terms against the binade constants (same expressions, same order)
static bool classify(const float* t, float M, float small, float hu, float& a)
{
  float r[4], d[4];
  for (int c = 0; c < 4; ++c)
  {
    volatile float m = M + t[c];
    r[c] = m - M;
    volatile float e = t[c] - r[c];
    d[c] = std::fabs(e);
  }
  const float dm = std::fmax(std::fmax(d[0], d[1]), std::fmax(d[2], d[3]));
  uint32_t mx = 0;
  for (int c = 0; c < 4; ++c)
    mx = std::max(mx, static_cast<uint32_t>(f2i(t[c])));
  volatile float x = r[0] + r[1];
  x = x + r[2];
  x = x + r[3];
  a = x;
  return (mx < static_cast<uint32_t>(f2i(small))) & (dm < hu);
}

static float emul(const std::vector<float>& terms, float s0)
{
  const int n = static_cast<int>(terms.size());
  const int n4 = (n + 3) >> 2;
  std::vector<float> row(static_cast<size_t>(n4) * 4 + 8, 0.0f);
  for (int i = 0; i < n; ++i)
    row[i] = terms[i];
  float s = s0;
  int q = 0;
  const int head = n4 < 64 ? n4 : 64;
  for (; q < head; ++q)
    s = quad(s, &row[4 * q]);
  while (q < n4)
  {
    const int sb = f2i(s) & 0x7f800000;
    if (f2i(s) < (40 << 23) || sb >= (254 << 23))
    {
      s = quad(s, &row[4 * q]);
      ++q;
      ++g_serial_quads;
      continue;
    }
    const float top = i2f(sb + (1 << 23));
    ++g_passes;
    const float M = i2f(sb | 0x00400000), small = i2f(sb - (1 << 23)), hu = i2f(sb - (24 << 23));
    float a[64];
    bool ok[64];
    for (int l = 0; l < 64; ++l)
    {
      const int mine = q + 2 * l;
      float ta[4] = { 0, 0, 0, 0 }, tb[4] = { 0, 0, 0, 0 };
      for (int c = 0; c < 4; ++c)
      {
        if (mine < n4)
          ta[c] = row[4 * mine + c];
        if (mine + 1 < n4)
          tb[c] = row[4 * (mine + 1) + c];
      }
      float a0, a1;
      const bool o0 = classify(ta, M, small, hu, a0), o1 = classify(tb, M, small, hu, a1);
      volatile float x = a0 + a1;
      a[l] = x;
      ok[l] = o0 & o1;
    }
    scan64(a);
    int L = -1;
    for (int l = 0; l < 64; ++l)
    {
      volatile float sp = s + a[l];
      if (!(ok[l] & (sp < top)))
      {
        L = l;
        break;
      }
    }
    if (L < 0)
    {
      s = s + a[63];
      q += 128;
      ++g_clean_passes;
      continue;
    }
    if (L > 0)
      s = s + a[L - 1];
    int at = q + 2 * L;
    at = at < n4 ? at : n4 - 1;
    s = quad(s, &row[4 * at]);
    s = quad(s, &row[4 * (at + 1)]);
    g_serial_quads += 2;
    q = at + 2;
  }
  return s;
}

static int check(const std::vector<float>& t, float s0, const char* what, long id)
{
  const float a = serial(t, s0), b = emul(t, s0);
  if (f2i(a) != f2i(b) && !(a != a && b != b))
  {
    printf("MISMATCH %s #%ld n=%zu s0=%g: serial %.9g (%08x) emul %.9g (%08x)\n", what, id, t.size(), s0, a, f2i(a), b, f2i(b));
    return 1;
  }
  return 0;
}

int main(int argc, char** argv)
{
  const long rounds = argc > 1 ? atol(argv[1]) : 300;
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  int bad = 0;
  for (long it = 0; it < rounds && bad < 10; ++it)
  {
    const int n = 1 + static_cast<int>(rng() % (it % 7 == 0 ? 70000 : 5000));
    std::vector<float> t(n);
    // 1. likelihood-like terms: (r - max(d, flat)) * w with many clamped (identical) terms and many zeros
    for (int i = 0; i < n; ++i)
    {
      const float d = 0.25f * U(rng);
      float dist = 0.2f - (d > 0.05f ? d : 0.05f);
      t[i] = (dist < 0.0f || (rng() & 3) == 0) ? 0.0f : dist * 5.0f;
    }
    bad += check(t, 0.0f, "likelihood-like", it);
    // 2. ties on purpose: terms that are odd multiples of half an ulp of the running sum's binade
    {
      float s = 0.0f;
      for (int i = 0; i < n; ++i)
      {
        if (s > 4.0f && (rng() % 3) == 0)
        {
          const int sb = f2i(s) & 0x7f800000;
          const float hu = i2f(sb - (24 << 23));
          t[i] = hu * static_cast<float>(2 * (rng() % 2000) + 1);
        }
        else
          t[i] = U(rng);
        s += t[i];
      }
    }
    bad += check(t, 0.0f, "ties", it);
    // 3. wild magnitudes (crossings everywhere, huge terms, denormals, zeros)
    for (int i = 0; i < n; ++i)
    {
      const int e = static_cast<int>(rng() % 60) - 45;
      t[i] = (rng() % 11 == 0) ? 0.0f : std::ldexp(U(rng), e);
      if (rng() % 997 == 0)
        t[i] = std::ldexp(1.0f, -140);
    }
    bad += check(t, 0.0f, "wild", it);
    // 4. equal terms (systematic rounding), a carried-in sum
    {
      const float c = (it & 1) ? 0.75f : 0.1f * (1 + it % 9);
      for (int i = 0; i < n; ++i)
        t[i] = c;
      bad += check(t, U(rng) * 100.0f, "equal", it);
    }
    // 5. signs, NaN / inf rarely (the serial fallback)
    for (int i = 0; i < n; ++i)
    {
      t[i] = U(rng) - ((it % 3 == 0) ? 0.3f : 0.0f);
      if (it % 50 == 49 && i == n / 2)
        t[i] = (it % 100 == 49) ? INFINITY : NAN;
    }
    bad += check(t, 0.0f, "signs", it);
    // 6. weights-like: n equal-magnitude small numbers (pf.h:255-260)
    for (int i = 0; i < n; ++i)
      t[i] = (1.0f / n) * (0.5f + U(rng)) * 30.0f;
    bad += check(t, 0.0f, "weights", it);
  }
  printf("float_chain_emul: %s; passes %ld (clean %ld), serial quads %ld\n", bad ? "FAILED" : "all equal", g_passes, g_clean_passes,
         g_serial_quads);
  return bad ? 1 : 0;
}
