// mcl3dl_hip.hip — the one translation unit behind the C ABI of include/mcl3dl_hip.h. Host side in the host_*.h and
// api_*.inl pieces included below (in this order), device code in kernels.h (likelihood / beam / pf / map compiler).
// gfx950 only; there is no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <limits>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>  // radix_sort_pairs for whole-map arrays only (host_cloud.h:radix_sort)

#include "kernels.h"
#include "mcl3dl_hip.h"

#pragma clang fp contract(off)

using namespace mcl3dl;

#include "host_context.h"
#include "host_map_compilers.h"
#include "host_measure.h"
#include "host_cloud.h"
#include "host_grid_builders.h"
#include "host_group.h"

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C"
{
#include "api_core.inl"
#include "api_reductions.inl"
#include "api_resample.inl"
#include "api_support.inl"
#include "api_cloud.inl"
#include "api_group.inl"
#include "api_group_state.inl"
}  // extern "C"
