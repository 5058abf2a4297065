// valu_microbench.hip — what one wave64 VALU instruction costs on a gfx950 SIMD, measured, so that bench.py's
// `roofline.valu_issue` fraction and the SQ_ACTIVE_INST_VALU counter can be read without guessing
// ("SQ_ACTIVE_INST_VALU 1.99e8 quad-cycles would read as 93 % if a wave64 f32 op held the pipe 4 cycles ...
// settle it with a micro-benchmark").
//
// Every kernel: ONE work-group of 256 x W threads per CU (two of 1024 threads for W = 8), so that each SIMD holds exactly
// W wavefronts; each wavefront issues ITER x 64 instructions of one kind, either as 8 independent chains or as one
// dependent chain, between two s_memtime reads (shader cycles). Output per (instruction, chains, W):
//   cycles per wave-instruction per SIMD = elapsed cycles / (instructions per wavefront x W)
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_microbench.bin valu_microbench.hip    Run: ./valu_microbench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
  do                                                                                          \
  {                                                                                           \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess)                                                                     \
    {                                                                                         \
      fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__);    \
      exit(1);                                                                                \
    }                                                                                         \
  } while (0)

constexpr int ITER = 2048;

// 8 instructions per macro use; INDEP = 8 different destination registers, DEP = the same one
#define OP8_INDEP(INS)                                                                                               \
  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n"   \
               INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8"                                              \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
               : "v"(b))
#define OP8_DEP(INS)                                                                                                 \
  asm volatile(INS " %0, %0, %1\n" INS " %0, %0, %1\n" INS " %0, %0, %1\n" INS " %0, %0, %1\n" INS " %0, %0, %1\n"   \
               INS " %0, %0, %1\n" INS " %0, %0, %1\n" INS " %0, %0, %1"                                              \
               : "+v"(a0)                                                                                             \
               : "v"(b))

#define KERNEL32(NAME, INS)                                                                    \
  __global__ __launch_bounds__(1024) void NAME##_indep(float* out, long long* cyc)              \
  {                                                                                            \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
          a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001f;                                            \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS);                          \
      OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS);                          \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }                                                                                            \
  __global__ __launch_bounds__(1024) void NAME##_dep(float* out, long long* cyc)                \
  {                                                                                            \
    float a0 = threadIdx.x, b = 1.0000001f;                                                    \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS);                                  \
      OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS);                                  \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;                                                  \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }

// 64-bit register operands (packed f32 pairs, f64)
#define KERNEL64(NAME, INS, TYPE)                                                              \
  __global__ __launch_bounds__(1024) void NAME##_indep(float* out, long long* cyc)              \
  {                                                                                            \
    TYPE a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,    \
         a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001;                                              \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS);                          \
      OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS); OP8_INDEP(INS);                          \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<float>(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }                                                                                            \
  __global__ __launch_bounds__(1024) void NAME##_dep(float* out, long long* cyc)                \
  {                                                                                            \
    TYPE a0 = threadIdx.x, b = 1.0000001;                                                      \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS);                                  \
      OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS); OP8_DEP(INS);                                  \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<float>(a0);                              \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }

KERNEL32(mul_f32, "v_mul_f32")
KERNEL32(add_f32, "v_add_f32")
KERNEL32(mul_lo_u32, "v_mul_lo_u32")
KERNEL32(mul_u32_u24, "v_mul_u32_u24")
KERNEL32(max_f32, "v_max_f32")
KERNEL64(add_f64, "v_add_f64", double)
KERNEL64(mul_f64, "v_mul_f64", double)

// packed f32: two floats in a 64-bit register pair
struct F2
{
  float x, y;
};
__global__ __launch_bounds__(1024) void pk_mul_f32_indep(float* out, long long* cyc)
{
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a0 = { 1.f + threadIdx.x, 2.f }, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
     a6 = a0 + 6.f, a7 = a0 + 7.f, b = { 1.0000001f, 0.9999999f };
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i)
  {
    OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32");
    OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32"); OP8_INDEP("v_pk_mul_f32");
  }
  const long long t1 = __builtin_readcyclecounter();
  const v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(1024) void pk_mul_f32_dep(float* out, long long* cyc)
{
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 a0 = { 1.f + threadIdx.x, 2.f }, b = { 1.0000001f, 0.9999999f };
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < ITER; ++i)
  {
    OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32");
    OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32"); OP8_DEP("v_pk_mul_f32");
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y;
  if ((threadIdx.x & 63) == 0)
    cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// unary transcendental / conversion (one source)
#define UOP8_INDEP(INS)                                                                                              \
  asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n"      \
               INS " %6, %6\n" INS " %7, %7"                                                                         \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define UKERNEL(NAME, INS)                                                                     \
  __global__ __launch_bounds__(1024) void NAME##_indep(float* out, long long* cyc)              \
  {                                                                                            \
    float a0 = 1.f + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,          \
          a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                               \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      UOP8_INDEP(INS); UOP8_INDEP(INS); UOP8_INDEP(INS); UOP8_INDEP(INS);                      \
      UOP8_INDEP(INS); UOP8_INDEP(INS); UOP8_INDEP(INS); UOP8_INDEP(INS);                      \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }
UKERNEL(sqrt_f32, "v_sqrt_f32")
UKERNEL(rcp_f32, "v_rcp_f32")
UKERNEL(cvt_flr_i32_f32, "v_cvt_flr_i32_f32")


// other VOP2 / VOP3 forms the likelihood kernel is made of
KERNEL32(sub_f32, "v_sub_f32")
KERNEL32(min_f32, "v_min_f32")
KERNEL32(and_b32, "v_and_b32")
KERNEL32(or_b32, "v_or_b32")
KERNEL32(lshlrev_b32, "v_lshlrev_b32")
KERNEL32(ashrrev_i32, "v_ashrrev_i32")
KERNEL32(add_u32, "v_add_u32")

#define XOP8(A, B)                                                                                                   \
  asm volatile(A "%0" B "\n" A "%1" B "\n" A "%2" B "\n" A "%3" B "\n" A "%4" B "\n" A "%5" B "\n" A "%6" B "\n" A "%7" B \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                      \
               : "v"(b)                                                                                               \
               : "vcc")
#define XKERNEL(NAME, A, B)                                                                    \
  __global__ __launch_bounds__(1024) void NAME##_indep(float* out, long long* cyc)              \
  {                                                                                            \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,   \
          a6 = a0 + 6, a7 = a0 + 7, b = 1.0000001f;                                            \
    const long long t0 = __builtin_readcyclecounter();                                         \
    for (int i = 0; i < ITER; ++i)                                                             \
    {                                                                                          \
      XOP8(A, B); XOP8(A, B); XOP8(A, B); XOP8(A, B);                                          \
      XOP8(A, B); XOP8(A, B); XOP8(A, B); XOP8(A, B);                                          \
    }                                                                                          \
    const long long t1 = __builtin_readcyclecounter();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
    if ((threadIdx.x & 63) == 0)                                                               \
      cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;                                      \
  }
// A "%d" B : the text before / after the destination register
XKERNEL(fma_f32, "v_fma_f32 ", ", %8, %8, %8")              // d = b*b + b  (no dependence on d: pure issue rate)
XKERNEL(fma_f32_acc, "v_fmac_f32 ", ", %8, %8")             // d += b*b
XKERNEL(mov_b32, "v_mov_b32 ", ", %8")
XKERNEL(cndmask_b32, "v_cndmask_b32 ", ", %8, %8, vcc")
XKERNEL(cmp_lt_f32, "v_cmp_lt_f32 vcc, ", ", %8")           // reads the register, writes vcc
XKERNEL(lshl_or_b32, "v_lshl_or_b32 ", ", %8, 3, %8")
XKERNEL(add3_u32, "v_add3_u32 ", ", %8, %8, %8")
XKERNEL(mul_f32_sgpr, "v_mul_f32 ", ", s4, %8")             // one SGPR source
XKERNEL(cvt_i32_f32, "v_cvt_i32_f32 ", ", %8")
XKERNEL(floor_f32, "v_floor_f32 ", ", %8")

typedef void (*kern_t)(float*, long long*);
struct Entry
{
  const char* name;
  kern_t k;
  int per_wave;  // instructions per wavefront
};

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out;
  long long* cyc;
  CHECK(hipMalloc(&out, sizeof(float) * 1024 * cus * 2));
  CHECK(hipMalloc(&cyc, sizeof(long long) * 16 * cus * 2));
  const int per = ITER * 64;
  const Entry entries[] = {
    { "v_mul_f32 x8 independent", mul_f32_indep, per },   { "v_mul_f32 dependent", mul_f32_dep, per },
    { "v_add_f32 x8 independent", add_f32_indep, per },   { "v_add_f32 dependent", add_f32_dep, per },
    { "v_max_f32 x8 independent", max_f32_indep, per },   { "v_pk_mul_f32 x8 independent", pk_mul_f32_indep, per },
    { "v_pk_mul_f32 dependent", pk_mul_f32_dep, per },    { "v_mul_u32_u24 x8 independent", mul_u32_u24_indep, per },
    { "v_mul_lo_u32 x8 independent", mul_lo_u32_indep, per }, { "v_add_f64 x8 independent", add_f64_indep, per },
    { "v_add_f64 dependent", add_f64_dep, per },          { "v_mul_f64 x8 independent", mul_f64_indep, per },
    { "v_sqrt_f32 x8 independent", sqrt_f32_indep, per }, { "v_rcp_f32 x8 independent", rcp_f32_indep, per },
    { "v_cvt_flr_i32_f32 x8 independent", cvt_flr_i32_f32_indep, per },
    { "v_sub_f32 x8 independent", sub_f32_indep, per },     { "v_min_f32 x8 independent", min_f32_indep, per },
    { "v_fma_f32 x8 independent", fma_f32_indep, per },     { "v_fmac_f32 x8 independent", fma_f32_acc_indep, per },
    { "v_mul_f32 (sgpr src) x8 independent", mul_f32_sgpr_indep, per },
    { "v_mov_b32 x8 independent", mov_b32_indep, per },     { "v_cndmask_b32 x8 independent", cndmask_b32_indep, per },
    { "v_cmp_lt_f32 x8 independent", cmp_lt_f32_indep, per }, { "v_and_b32 x8 independent", and_b32_indep, per },
    { "v_or_b32 x8 independent", or_b32_indep, per },       { "v_lshlrev_b32 x8 independent", lshlrev_b32_indep, per },
    { "v_ashrrev_i32 x8 independent", ashrrev_i32_indep, per }, { "v_add_u32 x8 independent", add_u32_indep, per },
    { "v_lshl_or_b32 x8 independent", lshl_or_b32_indep, per }, { "v_add3_u32 x8 independent", add3_u32_indep, per },
    { "v_cvt_i32_f32 x8 independent", cvt_i32_f32_indep, per }, { "v_floor_f32 x8 independent", floor_f32_indep, per },
  };
  printf("device: %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
  printf("%-36s %5s %14s %12s %12s %14s\n", "instruction", "W", "tick/instr/SIMD", "kernel ms", "ticks/ns", "cyc@2.4GHz");
  for (const Entry& e : entries)
    for (int W : { 1, 2, 4, 8 })
    {
      // W wavefronts per SIMD by construction: a work-group's wavefronts go round the CU's four SIMDs, so one work-group
      // of 256 x W threads per CU (two of 1024 for W = 8) puts exactly W on each — grids of 256-thread groups are NOT
      // spread evenly by the dispatcher (the first version of this benchmark measured that instead)
      const int threads = 256 * (W < 4 ? W : 4);
      const int blocks = cus * (W <= 4 ? 1 : W / 4);
      hipEvent_t a, b;
      CHECK(hipEventCreate(&a));
      CHECK(hipEventCreate(&b));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, cyc);  // warm-up
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, out, cyc);
      CHECK(hipEventRecord(b));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, a, b));
      std::vector<long long> h(static_cast<size_t>(threads / 64) * blocks);
      CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      double mean = 0;
      for (long long v : h)
        mean += static_cast<double>(v);
      mean /= h.size();
      // s_memtime counts at a fixed 100 MHz on some parts and shader cycles on others: report both views
      const double cyc_per = mean / (static_cast<double>(e.per_wave) * W);
      const double ghz = static_cast<double>(e.per_wave) * W * cyc_per / (ms * 1e6);
      printf("%-36s %5d %14.3f %12.4f %12.3f %14.3f\n", e.name, W, cyc_per, ms, ghz,
             ms * 2.4e6 / (static_cast<double>(e.per_wave) * W));
      CHECK(hipEventDestroy(a));
      CHECK(hipEventDestroy(b));
    }
  printf("# tick/instr/SIMD = mean per-wavefront elapsed s_memtime ticks / (instructions per wavefront x W wavefronts per SIMD);\n"
         "# ticks/ns = tick rate of that counter over the kernel (the shader clock if ~2.4, a fixed reference clock otherwise);\n"
         "# cyc@2.4GHz = hipEvent kernel time x 2.4 GHz / (instructions per wavefront x W): cycles per wave-instruction per SIMD\n"
         "# with no assumption about the counter (includes ~2 us of launch overhead).\n");
  return 0;
}
