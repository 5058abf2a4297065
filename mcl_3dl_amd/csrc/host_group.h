// host_group.h — part of the single translation unit mcl3dl_hip.hip: a device group = N contexts (one per GPU) owned by ONE
// host process, the in-process form of SURVEY.md §8e behind the C ABI (the reference node is one C++ process,
// src/mcl_3dl.cpp:1466).  Contiguous particle shards, map + scan replicated, one worker thread per device so that the
// launches of the N GPUs are enqueued concurrently, and ONE collective per update: ncclAllReduce(sum) of 2 + 2N doubles
// over RCCL (librccl is dlopen'ed the first time a group with more than one device needs it — a single-GPU user never
// loads it), or a host-side combine of the same record (collective = "host": used when device ids repeat — several
// contexts on one GPU, which RCCL refuses — and as a fallback when librccl is not installed).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enum values only: every RCCL call goes through the dlopen'ed pointers below

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace
{
struct RcclApi
{
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;   // optional: tears a communicator down with a collective still pending
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;

  // "" on success, otherwise why RCCL is not usable
  std::string load()
  {
    if (lib)
      return "";
    const char* env = getenv("MCL3DL_HIP_RCCL_LIB");
    const char* names[] = { env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    // a copy that is already in the process (e.g. the one torch brought) is preferred over loading a second one
    for (const char* n : names)
      if (n && !lib)
        lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (n && !lib)
        lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!lib)
      return std::string("librccl not found (") + dlerror() + ")";
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
    AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
    CommAbort = reinterpret_cast<decltype(CommAbort)>(dlsym(lib, "ncclCommAbort"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !AllGather || !GetErrorString)
    {
      lib = nullptr;
      return "librccl lacks ncclCommInitAll / ncclAllReduce / ncclAllGather";
    }
    return "";
  }
};

// One worker thread per device; run_all() hands every worker the same closure (argument = rank) and waits for all.
// Hand-over and completion are SPUN on for a short while before anybody sleeps on a condition variable: updates follow
// each other within a fraction of a millisecond while a filter runs, and a futex wake-up costs 20 - 50 us per hop — two hops
// per update, a tenth of a C2 update. A worker that has seen no task for SPIN_US goes to sleep (an idle filter burns nothing);
// the caller spins while the ranks work (that is its only job) and sleeps only behind a long update.
class WorkerPool
{
public:
  void start(int n)
  {
    rc_.assign(n, 0);
    for (int r = 0; r < n; ++r)
      threads_.emplace_back([this, r] { loop(r); });
  }
  void stop()
  {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_.store(true, std::memory_order_release);
    }
    cv_task_.notify_all();
    for (std::thread& t : threads_)
      t.join();
    threads_.clear();
  }
  // returns the first non-zero return code (by rank), 0 if every rank succeeded
  int run_all(const std::function<int(int)>& f, int* failed_rank)
  {
    if (threads_.empty())
    {
      const int rc = f(0);
      if (rc != 0 && failed_rank)
        *failed_rank = 0;
      return rc;
    }
    task_ = &f;
    pending_.store(static_cast<int>(threads_.size()), std::memory_order_relaxed);
    gen_.fetch_add(1);  // publishes task_ and pending_ (sequentially consistent: paired with the sleeper's count-then-look)
    if (sleepers_.load() > 0)
    {
      std::lock_guard<std::mutex> lk(m_);  // (a worker between its last look at gen_ and its wait holds m_)
      cv_task_.notify_all();
    }
    // completion: spin, then sleep
    const auto t0 = std::chrono::steady_clock::now();
    bool done = false;
    for (long spin = 0; !(done = pending_.load(std::memory_order_acquire) == 0); ++spin)
    {
      cpu_relax();
      if ((spin & 1023) == 1023 &&
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 2000.0)
        break;
    }
    if (!done)
    {
      std::unique_lock<std::mutex> lk(m_);
      waiter_.store(true);  // (sequentially consistent, like the worker's decrement-then-look: one of the two sees the other)
      cv_done_.wait(lk, [this] { return pending_.load() == 0; });
      waiter_.store(false);
    }
    task_ = nullptr;
    for (size_t r = 0; r < rc_.size(); ++r)
      if (rc_[r] != 0)
      {
        if (failed_rank)
          *failed_rank = static_cast<int>(r);
        return rc_[r];
      }
    return 0;
  }

private:
  static constexpr double SPIN_US = 300.0;
  void loop(int r)
  {
    uint64_t seen = 0;
    for (;;)
    {
      // the next task: spin for a while, then sleep
      const auto t0 = std::chrono::steady_clock::now();
      bool have = false;
      for (long spin = 0; !stop_.load(std::memory_order_acquire); ++spin)
      {
        if (gen_.load(std::memory_order_acquire) != seen)
        {
          have = true;
          break;
        }
        cpu_relax();
        if ((spin & 255) == 255 &&
            std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > SPIN_US)
          break;
      }
      if (!have)
      {
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1);
        cv_task_.wait(lk, [&] { return stop_.load() || gen_.load() != seen; });
        sleepers_.fetch_sub(1);
      }
      if (stop_.load(std::memory_order_acquire))
        return;
      seen = gen_.load(std::memory_order_acquire);
      const std::function<int(int)>* f = task_;
      rc_[r] = (*f)(r);
      if (pending_.fetch_sub(1) == 1 && waiter_.load())
      {
        std::lock_guard<std::mutex> lk(m_);
        cv_done_.notify_one();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_task_, cv_done_;
  const std::function<int(int)>* task_ = nullptr;
  std::atomic<uint64_t> gen_{ 0 };
  std::atomic<int> pending_{ 0 }, sleepers_{ 0 };
  std::atomic<bool> stop_{ false }, waiter_{ false };
  std::vector<int> rc_;
};

// Every worker of one run_all() casts one vote and gets the conjunction of all votes back. The update's collective is
// enqueued only when every rank reached it: a rank whose upload / launch failed must not leave the others blocked inside
// an all-reduce that can never complete (api_group.inl, the RCCL path).
class VoteBarrier
{
public:
  void resize(int n)
  {
    n_ = n;
  }
  bool vote(bool ok)
  {
    if (n_ <= 1)
      return ok;
    std::unique_lock<std::mutex> lk(m_);
    const uint64_t gen = gen_;
    if (count_ == 0)
      all_ok_ = true;
    all_ok_ = all_ok_ && ok;
    if (++count_ == n_)
    {
      result_[gen & 1] = all_ok_;
      count_ = 0;
      ++gen_;
      cv_.notify_all();
      return result_[gen & 1];
    }
    cv_.wait(lk, [&] { return gen_ != gen; });
    return result_[gen & 1];
  }

private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_ = 1, count_ = 0;
  uint64_t gen_ = 0;
  bool all_ok_ = true, result_[2] = { true, true };
};
}  // namespace

struct mcl3dl_hip_group
{
  std::vector<mcl3dl_hip_ctx*> ctx;
  std::vector<int> devices;
  std::string err;
  int collective = 0;  // 0 = RCCL all-reduce, 1 = host combine
  int direct_single = 1;  // 1 = a group of one device calls its context directly; 0 = it takes the sharded path too
  bool devices_distinct = true;
  WorkerPool pool;
  VoteBarrier vote;
  int inject_failure_rank = -1;  // test hook (option "inject_failure_rank"): that rank fails ahead of the collective, once
  OrderedScan scan;  // ordered once per update, pushed to every device
  // RCCL (created on first use by a group of more than one device)
  RcclApi rccl;
  std::vector<ncclComm_t> comms;
  uint64_t collectives_rccl = 0, collectives_host = 0;
  // per-rank scratch of the host combine
  std::vector<std::vector<double>> host_packed;
  // poses kept on the devices by group_upload_poses
  size_t n_pose_uploaded = 0;
  // the batch mcl3dl_hip_group_measure_batch_begin started
  size_t prog_n_p = 0;
  bool prog_direct = false;
  bool prog_sharded = false;          // every rank holds a progressive batch of its own over its shard (api_group.inl)
  std::vector<bool> prog_done;        // ... ranks whose whole shard has been handed to the caller's arrays
  // particles resident on the devices (api_group_state.inl): 13-float states + weights, sharded by shard_bounds
  size_t n_resident = 0;
  std::vector<float> h_weight, h_state;  // host gather buffers of the resampling steps
  std::vector<double> h_parts;           // per-rank records of the reductions
  size_t rs_n_out = 0;                   // set by group_resample_begin
  bool rs_begun = false, rs_planned = false;
  size_t rs_n_dup = 0;

  // a communicator that saw a failed or abandoned collective: aborted where RCCL can (a pending collective would block a
  // plain destroy), rebuilt by group_comms on next use
  void drop_comms()
  {
    for (ncclComm_t& c : comms)
      if (c)
      {
        if (rccl.CommAbort)
          (void)rccl.CommAbort(c);
        else
          (void)rccl.CommDestroy(c);
        c = nullptr;
      }
    comms.clear();
  }

  int n() const
  {
    return static_cast<int>(ctx.size());
  }
  int fail(int code, const char* fmt, ...)
  {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  int fail_rank(int rc, int r)
  {
    return fail(rc, "device %d (rank %d): %s", devices[r], r, ctx[r]->err.c_str());
  }
};

namespace
{
// contiguous shard [lo, hi) of rank r when n particles are split over `world` ranks (sizes differ by at most one):
// the same rule as mcl_3dl_amd/distributed.py:shard_bounds
inline void shard_bounds(size_t n, int world, int r, size_t* lo, size_t* hi)
{
  const size_t base = n / world, rem = n % world;
  *lo = r * base + std::min<size_t>(r, rem);
  *hi = *lo + base + (static_cast<size_t>(r) < rem ? 1 : 0);
}
}  // namespace
