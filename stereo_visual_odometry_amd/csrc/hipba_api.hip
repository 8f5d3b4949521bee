// hipba_api.hip -- the C ABI of include/hipba.h: problem analysis on the host, residency
// management and the host side of the Levenberg-Marquardt loop.
//
// The host does what ceres::Problem / Program preprocessing does for the reference
// (src/solve/optimizer.cpp:16-52): it drops constant and unobserved blocks, classifies
// residual blocks, and lays everything out as flat arrays.  The per-iteration arithmetic
// and the accept/reject logic run in lm_kernels.hip; the host only queues kernels and
// polls one integer (the number of windows still iterating) per iteration.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <system_error>
#include <thread>
#include <limits>
#include <vector>

#include "camera_order.hpp"
#include "hipba_internal.hpp"

using namespace hipba;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIPBA_HIP(expr)                                                                               \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return fail(e__ == hipErrorOutOfMemory ? HIPBA_ERR_OUT_OF_MEMORY : HIPBA_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                  hipGetErrorString(e__), __FILE__, __LINE__);                                        \
  } while (0)

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// A few host threads kept by a handle for the analysis of small windows: that analysis takes a few hundred
// microseconds, of which starting and joining fresh threads was a quarter.  run(n, f) calls f(0) .. f(n - 1), f(0) on
// the caller; the workers sleep on a condition variable in between.
class HostWorkers {
 public:
  ~HostWorkers() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_ = true;
    }
    go_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // returns false if a part threw (out of memory): the exception stays inside its thread
  bool run(int n, const std::function<void(int)>& f) {
    failed_ = false;
    if (n <= 1) { if (n == 1) call(f, 0); return !failed_; }
    while ((int)threads_.size() < n - 1) {
      const int id = (int)threads_.size() + 1;
      threads_.emplace_back([this, id] { loop(id); });
    }
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &f; parts_ = n; pending_ = n - 1; ++generation_;
    }
    go_.notify_all();
    call(f, 0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
    return !failed_;
  }
  // f(1) on a worker while the caller goes on; f must stay alive until wait() has returned, and no other job may be
  // handed out in between.
  void start(const std::function<void(int)>& f) {
    failed_ = false;
    if (threads_.empty()) threads_.emplace_back([this] { loop(1); });
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &f; parts_ = 2; pending_ = 1; ++generation_;
    }
    go_.notify_all();
  }
  bool wait() {
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
    return !failed_;
  }

 private:
  void call(const std::function<void(int)>& f, int i) {
    try { f(i); } catch (...) { failed_ = true; }
  }
  void loop(int id) {
    long long seen = 0;
    for (;;) {
      const std::function<void(int)>* job = nullptr;
      {
        std::unique_lock<std::mutex> lk(m_);
        go_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
        if (id < parts_) job = job_;
      }
      if (job) {
        call(*job, id);
        std::lock_guard<std::mutex> lk(m_);
        if (--pending_ == 0) done_.notify_one();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable go_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int parts_ = 0, pending_ = 0;
  long long generation_ = 0;
  bool stop_ = false;
  std::atomic<bool> failed_{false};
};

// ---- scratch of hipba_evaluate_device ---------------------------------------------------------------
// That entry point serves callers that cannot allocate, so the one buffer its kernels need beyond the caller's
// inputs and outputs -- the per-workgroup partial sums of the cost, one double per 256 observations -- lives in the
// code object itself.  Every GPU the library runs on holds its own copy (512 KB) until the process ends.  In host
// code the NAME of the array is not a device address: the address on a device comes from hipGetSymbolAddress, asked
// once per device with that device current (scratch_on_device, called by hipba_create), and only that pointer is
// ever handed to a kernel.  Nothing in it has to start at zero: a call overwrites every word it reads.
static __device__ __align__(256) double scratch[HIPBA_EVALUATE_DEVICE_MAX_OBS / kBlock];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[kMaxDevices] = {};

// device address of `scratch` on `device`, which must be the current device
hipError_t scratch_on_device(int device, double** out) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(g_scratch_mutex);
  if (!g_scratch_ptr[device]) {
    double* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    g_scratch_ptr[device] = ptr;
  }
  *out = g_scratch_ptr[device];
  return hipSuccess;
}
}  // namespace

struct hipba_handle_s {
  int device = 0;
  double* eval_scratch = nullptr;  // this device's copy of `scratch` (see above)
  HostWorkers workers;
  hipStream_t stream = nullptr;
  // grow-only buffers reused by hipba_solve / hipba_solve_batched
  void* arena = nullptr;
  size_t arena_bytes = 0;
  void* staging = nullptr;  // pinned host memory
  size_t staging_bytes = 0;
  int* h_count = nullptr;   // pinned: windows still active, one word per slot of the poll ring
  hipEvent_t poll_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // ring of "activity count copied" events
  // hipba_options::report_time_breakdown: four timestamps per iteration (before / after the camera-block pass,
  // before / after the candidate cost pass), one set per slot of the poll ring; created on first use
  hipEvent_t phase_ev[4][4] = {};
  bool phase_ev_ready = false;
  // pipelined exchange of sharded solves (TileFeed): a stream for the chunks' all-reduces, "the tiles are packed" on
  // the engine's stream, "chunk c has arrived" on the exchange stream; created on first use
  hipStream_t xchg_stream = nullptr;
  hipEvent_t xchg_packed = nullptr;
  hipEvent_t xchg_ready[kMaxFeedChunks] = {};
  bool xchg_ready_all = false;  // the stream and every event above exist
};

struct WinMeta {
  int n_poses = 0, n_points = 0;
  long long n_obs = 0;
  int pose_base = 0;
  int n_act_poses = 0, n_act_points = 0;
  long long n_act_obs = 0;
  std::vector<int> dp_of;  // original landmark index -> device landmark index
};

struct hipba_device_problem_s {
  void* arena = nullptr;
  size_t arena_bytes = 0;
  bool owns_arena = false;
  DevP P{};
  std::vector<WinMeta> meta;
  double* poses0 = nullptr;   // uploaded parameters, kept for hipba_reset
  double* points0 = nullptr;
  int max_ncam = 0;
  int max_pm_blocks = 0;      // most PM blocks in one window
  long long upload_bytes = 0; // host-to-device bytes of the upload
  long long d2h_bytes = 0;    // device-to-host bytes since the last solve started
  int n_blocks = 0;
  long long S_doubles = 0;
  bool S_fully_written = false;  // gathered mode, every off-diagonal block has a pair: no word of S needs zeroing
  BigWork big;
  bool has_big = false;
  // first tile column of every 128-row tile of the reduced matrix (camera_order.hpp): what the dense
  // factorisation may skip.  A shard knows its own landmarks' couplings only: sharded solves merge these first.
  std::vector<int> env_first;
  // end-of-solve snapshot in the handle's pinned staging buffer (small one-shot solves): the control words and both
  // parameter buffers, fetched with the solve's own final synchronisation so that reading the result back costs no
  // further round trip over the host link
  bool tail_valid = false;
  size_t tail_span = 0;  // bytes from the start of the arena that hold both parameter buffers and the window states
  const WinState* tail_ws = nullptr;
  const double* tail_poses[2] = {nullptr, nullptr};
  const double* tail_points[2] = {nullptr, nullptr};
};

namespace {

// workgroups per window in the small-window landmark elimination: enough to fill the chip, few enough
// that the per-workgroup flush of the LDS accumulators stays a small part of the work
int schur_parts_for(int n_windows, int max_pm_blocks) {
  // development switch: HIPBA_SCHUR_WORKGROUPS=<n> sets the number of workgroups a launch aims for (default 4096)
  static const int target = [] { const char* e = getenv("HIPBA_SCHUR_WORKGROUPS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 4096; }();
  const int want = (target + n_windows - 1) / std::max(1, n_windows);
  return std::max(1, std::min(want, max_pm_blocks));
}

void free_resident(hipba_device_problem_s* R) {
  if (!R) return;
  if (R->owns_arena && R->arena) (void)hipFree(R->arena);
  delete R;
}
struct ResidentDeleter {
  void operator()(hipba_device_problem_s* R) const { free_resident(R); }
};

// Nothing may leave an extern "C" function by unwinding: every entry point runs its body through this.
template <typename F>
int guarded(F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail(HIPBA_ERR_OUT_OF_MEMORY, "host memory allocation failed");
  } catch (const std::system_error& e) {
    return fail(HIPBA_ERR_OUT_OF_MEMORY, "host resource exhausted: %s", e.what());
  } catch (const std::exception& e) {
    return fail(HIPBA_ERR_INTERNAL, "unexpected exception: %s", e.what());
  } catch (...) {
    return fail(HIPBA_ERR_INTERNAL, "unexpected exception");
  }
}

// ---- arena planning ----------------------------------------------------------------------------
struct ArenaPlan {
  struct Item { void** dst; size_t bytes; size_t off; };
  std::vector<Item> items;
  size_t total = 0;
  template <typename T>
  void add(T** dst, size_t count) {
    Item it;
    it.dst = reinterpret_cast<void**>(dst);
    it.bytes = count * sizeof(T);
    it.off = total;
    total = align_up(total + std::max<size_t>(it.bytes, 16), 256);
    items.push_back(it);
  }
  void bind(void* base) {
    for (auto& it : items) *it.dst = static_cast<char*>(base) + it.off;
  }
};

int ensure_arena(hipba_handle_s* h, size_t bytes) {
  if (h->arena_bytes >= bytes) return HIPBA_OK;
  if (h->arena) { HIPBA_HIP(hipFree(h->arena)); h->arena = nullptr; h->arena_bytes = 0; }
  // (never less than 16 MB: a sequence's first windows grow frame by frame, and every regrowth is a hipFree and a
  // hipMalloc -- a few hundred microseconds -- in the middle of a 1 ms call; a keyframe window needs 3-4 MB)
  const size_t want = std::max<size_t>(align_up(bytes + bytes / 4, 1 << 20), (size_t)16 << 20);
  HIPBA_HIP(hipMalloc(&h->arena, want));
  h->arena_bytes = want;
  return HIPBA_OK;
}

int ensure_staging(hipba_handle_s* h, size_t bytes) {
  if (h->staging_bytes >= bytes) return HIPBA_OK;
  if (h->staging) { HIPBA_HIP(hipHostFree(h->staging)); h->staging = nullptr; h->staging_bytes = 0; }
  const size_t want = std::max<size_t>(align_up(bytes + bytes / 4, 1 << 20), (size_t)8 << 20);  // (as the arena: not below 8 MB)
  HIPBA_HIP(hipHostMalloc(&h->staging, want, hipHostMallocDefault));
  h->staging_bytes = want;
  return HIPBA_OK;
}

LmParams make_lm(const hipba_options& o) {
  LmParams lm;
  lm.max_num_iterations = o.max_num_iterations;
  lm.loss = o.loss;
  lm.loss_scale = o.loss_scale;
  lm.jacobi_scaling = o.jacobi_scaling;
  lm.max_invalid = o.max_num_consecutive_invalid_steps;
  lm.max_radius = o.max_trust_region_radius;
  lm.min_radius = o.min_trust_region_radius;
  lm.min_relative_decrease = o.min_relative_decrease;
  lm.min_lm_diagonal = o.min_lm_diagonal;
  lm.max_lm_diagonal = o.max_lm_diagonal;
  lm.function_tolerance = o.function_tolerance;
  lm.gradient_tolerance = o.gradient_tolerance;
  lm.parameter_tolerance = o.parameter_tolerance;
  lm.initial_radius = o.initial_trust_region_radius;
  lm.param_tol_needs_success = o.parameter_tolerance_needs_success;
  return lm;
}

int validate_problem(const hipba_problem& p, int w, bool check_indices = true) {
  if (p.n_poses < 0 || p.n_points < 0 || p.n_obs < 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: negative size", w);
  if ((p.n_poses && !p.poses) || (p.n_points && !p.points))
    return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: null parameter array", w);
  if (p.n_obs && (!p.obs_pose || !p.obs_point || !p.obs_uv))
    return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: null observation array", w);
  if (p.camera_model != HIPBA_CAMERA_Z_FORWARD && p.camera_model != HIPBA_CAMERA_X_FORWARD)
    return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: unknown camera model %d", w, p.camera_model);
  for (long long k = 0; check_indices && k < p.n_obs; ++k) {
    if (p.obs_pose[k] < 0 || p.obs_pose[k] >= p.n_poses || p.obs_point[k] < 0 || p.obs_point[k] >= p.n_points)
      return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: observation %lld references pose %d / landmark %d out of range",
                  w, k, p.obs_pose[k], p.obs_point[k]);
  }
  return HIPBA_OK;
}

// development aid (-DHIPBA_SETUP_TIMING): host time between the marks of build_resident, to stderr
#ifdef HIPBA_SETUP_TIMING
struct SetupClock {
  double t = now_s();
  void mark(const char* what) { const double n = now_s(); fprintf(stderr, "setup %-16s %7.1f us\n", what, 1e6 * (n - t)); t = now_s(); }
};
#define SETUP_CLOCK() SetupClock setup_clock
#define SETUP_MARK(what) setup_clock.mark(what)
#else
#define SETUP_CLOCK() do {} while (0)
#define SETUP_MARK(what) do {} while (0)
#endif

// ---- analysis + upload -------------------------------------------------------------------------
// pose_order (one-window problems, optional): the poses of the window in the order their blocks should take in the
// reduced camera system (a permutation of 0 .. n_poses - 1; constant and unobserved ones are skipped).  Null: chosen
// here (see camera_order.hpp).  Shards of one problem must all be given the same order.
int build_resident(hipba_handle_s* h, int nw, const hipba_problem* probs, bool use_handle_arena, int log_cap,
                   bool all_poses_active, hipba_device_problem_s** out, const int32_t* pose_order = nullptr) {
  SETUP_CLOCK();
  for (int w = 0; w < nw; ++w) {
    int rc = validate_problem(probs[w], w, /*check_indices=*/false);  // (the indices: in the counting pass below)
    if (rc) return rc;
  }
  // owned here until handed to the caller: every early return and every exception releases it (and its arena)
  std::unique_ptr<hipba_device_problem_s, ResidentDeleter> owner(new hipba_device_problem_s());
  hipba_device_problem_s* R = owner.get();
  R->meta.resize(nw);
  DevP& P = R->P;
  P = DevP{};
  P.n_windows = nw;

  long long NP = 0, NPT = 0;
  for (int w = 0; w < nw; ++w) {
    R->meta[w].n_poses = probs[w].n_poses;
    R->meta[w].n_points = probs[w].n_points;
    R->meta[w].n_obs = probs[w].n_obs;
    R->meta[w].pose_base = (int)NP;
    NP += probs[w].n_poses; NPT += probs[w].n_points;
  }
  if (NP > 0x7fffffffLL / 8 || NPT > 0x7fffffffLL / 4) return fail(HIPBA_ERR_UNSUPPORTED, "problem too large for 32-bit block indices");

  // per landmark / pose bookkeeping
  std::vector<long long> pt_base(nw + 1, 0);
  for (int w = 0; w < nw; ++w) pt_base[w + 1] = pt_base[w] + probs[w].n_points;
  std::vector<int> deg(NPT, 0);          // observations of a variable landmark
  std::vector<int> pose_nobs(NP, 0);
  std::vector<uint8_t> pose_c(NP, 0), pt_c(NPT, 0);
  long long n_vc = 0, n_cc = 0;
  for (int w = 0; w < nw; ++w) {
    const hipba_problem& p = probs[w];
    for (int i = 0; i < p.n_poses; ++i) pose_c[R->meta[w].pose_base + i] = p.pose_const ? (p.pose_const[i] != 0) : 0;
    for (int i = 0; i < p.n_points; ++i) pt_c[pt_base[w] + i] = p.point_const ? (p.point_const[i] != 0) : 0;
    // one pass: the indices checked, observations per variable landmark and per pose, and the sizes of the two
    // sections of observations of constant landmarks (variable pose / constant pose)
    const unsigned np_w = (unsigned)p.n_poses, npt_w = (unsigned)p.n_points;
    const int pose_base = R->meta[w].pose_base;
    for (long long k = 0; k < p.n_obs; ++k) {
      const unsigned ip = (unsigned)p.obs_pose[k], iq = (unsigned)p.obs_point[k];
      if (ip >= np_w || iq >= npt_w)
        return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d: observation %lld references pose %d / landmark %d out of range",
                    w, k, p.obs_pose[k], p.obs_point[k]);
      const long long gp = pt_base[w] + iq;
      if (!pt_c[gp]) deg[gp]++;
      else if (pose_c[pose_base + ip]) ++n_cc;
      else ++n_vc;
      pose_nobs[pose_base + ip]++;
    }
  }
  // device landmark order: variable observed landmarks first (by window, index), the rest after
  std::vector<int> dp_of(NPT, -1);
  std::vector<WinDesc> wd(nw);
  int n_pm_points = 0;
  bool has_long = false;
  for (int w = 0; w < nw; ++w) {
    wd[w] = WinDesc{};
    wd[w].pp_begin = n_pm_points;
    for (int i = 0; i < probs[w].n_points; ++i) {
      const long long gp = pt_base[w] + i;
      if (deg[gp] > 0) {
        if (deg[gp] > kBlock) has_long = true;  // walks its observations block by block (see LongDesc)
        dp_of[gp] = n_pm_points++;
      }
    }
    wd[w].pp_end = n_pm_points;
    R->meta[w].n_act_points = wd[w].pp_end - wd[w].pp_begin;
  }
  {
    int next = n_pm_points;
    for (long long gp = 0; gp < NPT; ++gp)
      if (dp_of[gp] < 0) dp_of[gp] = next++;
  }
  // global slots of variable observed poses
  std::vector<int> pose_gslot(NP, -1), pose_window(NP, 0), slot_pose;
  int n_slots = 0, max_ncam = 0;
  long long S_doubles = 0;
  for (int w = 0; w < nw; ++w) {
    const hipba_problem& p = probs[w];
    wd[w].K = CameraK{p.fx, p.fy, p.cx, p.cy, p.camera_model};
    wd[w].pose_begin = R->meta[w].pose_base;
    wd[w].pose_end = wd[w].pose_begin + p.n_poses;
    wd[w].cam_base = n_slots;
    for (int i = 0; i < p.n_poses; ++i) {
      const int c = wd[w].pose_begin + i;
      pose_window[c] = w;
      // a shard keeps every variable pose so that all ranks agree on the layout of the reduced system
      if (!pose_c[c] && (pose_nobs[c] > 0 || all_poses_active)) { pose_gslot[c] = n_slots++; slot_pose.push_back(c); }
    }
    wd[w].ncam = n_slots - wd[w].cam_base;
    R->meta[w].n_act_poses = wd[w].ncam;
    max_ncam = std::max(max_ncam, wd[w].ncam);
    wd[w].S_off = S_doubles;
    S_doubles += 36ll * wd[w].ncam * (wd[w].ncam + 1) / 2;
    wd[w].log_base = w * log_cap;
  }
  if (max_ncam * 6 > kSmallMaxN && nw > 1) {
    return fail(HIPBA_ERR_UNSUPPORTED, "batched windows are limited to %d variable poses each (got %d)", kSmallMaxN / 6,
                max_ncam);
  }

  SETUP_MARK("counts");
  // ---- camera-major lists and the parameters in device order -----------------------------------------
  // Straight from the caller's arrays (every observation by a variable pose, in the caller's order), so they depend
  // on nothing below but the slots: for windows whose slots are final here (not the one large window whose poses are
  // reordered further down) a worker of the handle builds them beside the threads that sort the observations by
  // landmark.
  std::vector<long long> cm_start(n_slots + 1, 0);
  std::vector<int> cm_dp;
  std::vector<double> cm_uv;
  std::vector<double> poses(std::max<long long>(7 * NP, 7)), points(std::max<long long>(3 * NPT, 3));
  auto build_side_lists = [&]() {
    for (int g = 0; g < n_slots; ++g) cm_start[g + 1] = pose_nobs[slot_pose[g]];
    for (int g = 0; g < n_slots; ++g) cm_start[g + 1] += cm_start[g];
    const long long n_cm = cm_start[n_slots];
    cm_dp.resize(std::max<long long>(n_cm, 1));
    cm_uv.resize(std::max<long long>(2 * n_cm, 2));
    std::vector<long long> cur(cm_start.begin(), cm_start.end() - 1);
    for (int w = 0; w < nw; ++w) {
      const hipba_problem& p = probs[w];
      const int pose_base = R->meta[w].pose_base;
      for (long long k = 0; k < p.n_obs; ++k) {
        const int g = pose_gslot[pose_base + p.obs_pose[k]];
        if (g < 0) continue;
        const long long dst = cur[g]++;
        cm_dp[dst] = dp_of[pt_base[w] + p.obs_point[k]];
        cm_uv[2 * dst] = p.obs_uv[2 * k]; cm_uv[2 * dst + 1] = p.obs_uv[2 * k + 1];
      }
      if (p.n_poses) memcpy(poses.data() + 7ll * pose_base, p.poses, sizeof(double) * 7 * p.n_poses);
      for (int i = 0; i < p.n_points; ++i) {
        const int dp = dp_of[pt_base[w] + i];
        points[3ll * dp] = p.points[3 * i]; points[3ll * dp + 1] = p.points[3 * i + 1]; points[3ll * dp + 2] = p.points[3 * i + 2];
      }
    }
  };
  bool side_lists_done = false;
  std::atomic<bool> lists_claimed{false};
  const std::function<void(int)> lists_job = [&](int) { if (!lists_claimed.exchange(true)) build_side_lists(); };
  bool worker_busy = false;  // lists_job handed to a worker and not yet collected
  // (collects it on every way out of this function: the job refers to this frame)
  struct CollectWorker {
    HostWorkers& w; bool& busy;
    ~CollectWorker() { if (busy) (void)w.wait(); }
  } collect_worker{h->workers, worker_busy};
  // ---- observations: PM | VC | CC --------------------------------------------------------------
  std::vector<long long> pp_start(n_pm_points + 1, 0);
  for (long long gp = 0; gp < NPT; ++gp)
    if (deg[gp] > 0) pp_start[dp_of[gp] + 1] = deg[gp];
  for (int i = 0; i < n_pm_points; ++i) pp_start[i + 1] += pp_start[i];
  const long long n_pm_obs = pp_start[n_pm_points];
  const long long n_act_obs = n_pm_obs + n_vc, n_all = n_act_obs + n_cc;
  std::vector<int> obs_pose(std::max<long long>(n_all, 1)), obs_dp(std::max<long long>(n_all, 1));
  std::vector<double> obs_uv(std::max<long long>(2 * n_all, 2));
  std::vector<int> vc_window_count(nw, 0), cc_window_count(nw, 0);
  // Per variable landmark, gathered while its observations are dropped into place: how many of them are by variable
  // poses, and -- where the slots are final already (every case but the one large window, whose poses are reordered
  // below) -- whether its segment arrives sorted by (slot, pose) and free of repeats.  A caller that lists its
  // observations frame by frame (the reference does: optimizer.cpp:37-51) hands over sorted segments, and the two
  // passes over the observations that sort the segments and look for repeats are then skipped.
  const bool dense_path = nw == 1 && 6 * max_ncam > kSmallMaxN;
  // (what the placement touches per observation sits together: one cache line by landmark and one word by point instead
  // of three and two)
  struct SegState { long long cur, last_key; int nvar, pad; };
  std::vector<SegState> seg(std::max(n_pm_points, 1));
  for (int d = 0; d < n_pm_points; ++d) seg[d] = SegState{pp_start[d], std::numeric_limits<long long>::min(), 0, 0};
  std::vector<int> where(NPT);  // device landmark index of a point; of a constant one, its complement (negative)
  for (long long gp = 0; gp < NPT; ++gp) where[gp] = pt_c[gp] ? ~dp_of[gp] : dp_of[gp];
  std::vector<int> nvar(std::max(n_pm_points, 1), 0);
  bool seg_unsorted = false, seg_repeat = false;
  {
    // The landmarks can be cut into ranges, one per thread: every thread reads all observations and places those of its
    // landmarks (what it writes -- cursors, counts, the segments themselves -- is its own); thread 0 also places the
    // observations of constant landmarks.  Beside them one more thread builds the camera-major lists, which need
    // nothing from here.  Measured on a 25 000-observation window: a job on the handle's workers costs 17-18 us to
    // hand out and collect, and a part's time is its scan of ALL observations with a range test that the branch
    // predictor cannot learn (85 us each for three parts against 103 us for one thread doing everything) -- so a window
    // is placed by ONE thread with the lists beside it, and only inputs of a million observations are cut up.
    const int n_parts = n_all >= 1000000 ? 4 : 1;
    bool part_unsorted[4] = {false, false, false, false}, part_repeat[4] = {false, false, false, false};
    auto place = [&](int part) {
      const int d_lo = (int)((long long)n_pm_points * part / n_parts), d_hi = (int)((long long)n_pm_points * (part + 1) / n_parts);
      long long vc_cur = n_pm_obs, cc_cur = n_act_obs;
      bool unsorted = false, repeat = false;
      for (int w = 0; w < nw; ++w) {
        const hipba_problem& p = probs[w];
        const int pose_base = R->meta[w].pose_base;
        for (long long k = 0; k < p.n_obs; ++k) {
          const int wh = where[pt_base[w] + p.obs_point[k]];
          long long dst;
          int c, d;
          if (wh >= 0) {
            d = wh;
            if (d < d_lo || d >= d_hi) continue;
            c = pose_base + p.obs_pose[k];
            const int g = pose_gslot[c];
            SegState& sg = seg[d];
            dst = sg.cur++;
            sg.nvar += g >= 0;
            if (!dense_path) {
              const long long key = ((long long)g << 32) + c;
              if (key < sg.last_key) unsorted = true;
              else if (key == sg.last_key && g >= 0) repeat = true;
              sg.last_key = key;
            }
          } else {
            if (part != 0) continue;
            d = ~wh;
            c = pose_base + p.obs_pose[k];
            if (!pose_c[c]) { dst = vc_cur++; vc_window_count[w]++; }
            else { dst = cc_cur++; cc_window_count[w]++; }
          }
          obs_pose[dst] = c;
          obs_dp[dst] = d;
          obs_uv[2 * dst] = p.obs_uv[2 * k];
          obs_uv[2 * dst + 1] = p.obs_uv[2 * k + 1];
        }
      }
      part_unsorted[part] = unsorted; part_repeat[part] = repeat;
    };
    const bool lists_here = !dense_path && n_slots > 0;  // (the large window's slots are not final yet)
    const int n_jobs = n_parts + (lists_here && n_all >= 12000 ? 1 : 0);
    if (n_parts == 1) {
      // A window: the lists are offered to a worker and this thread places the observations meanwhile.  Whoever gets
      // to the lists first builds them -- a worker that wakes late (a sleeping core) finds them taken and costs
      // nothing, where waiting for it cost up to 0.2 ms now and then.
      if (n_jobs > 1) { h->workers.start(lists_job); worker_busy = true; }
      place(0);
      if (lists_here) {
        if (!lists_claimed.exchange(true)) build_side_lists();
        else if (!h->workers.wait()) { worker_busy = false; return fail(HIPBA_ERR_OUT_OF_MEMORY, "host analysis of the camera-major lists ran out of memory"); }
        else worker_busy = false;
        side_lists_done = true;
      }
    } else {
      if (!h->workers.run(n_jobs, [&](int i) { if (i < n_parts) place(i); else build_side_lists(); }))
        return fail(HIPBA_ERR_OUT_OF_MEMORY, "host analysis of the observations ran out of memory");
      side_lists_done = n_jobs > n_parts;
    }
    for (int i = 0; i < n_parts; ++i) { seg_unsorted |= part_unsorted[i]; seg_repeat |= part_repeat[i]; }
    for (int d = 0; d < n_pm_points; ++d) nvar[d] = seg[d].nvar;
    for (int w = 0; w < nw; ++w) R->meta[w].n_act_obs = 0;
  }

  SETUP_MARK("observations");
  // ---- order of the poses in the reduced system (one large window: the dense factorisation's envelope) --------
  const bool segments_known_clean = !dense_path && !seg_unsorted && !seg_repeat;  // sorted by slot, no pose twice
  if (dense_path) {
    const int ncam = n_slots;
    const int n_tiles = (6 * ncam + kCholNB - 1) / kCholNB;
    // co-visibility of the variable poses: two poses are coupled in S when they share a variable landmark
    PoseGraph graph(ncam);
    {
      std::vector<int> seg;
      for (int pp = 0; pp < n_pm_points; ++pp) {
        seg.clear();
        for (long long o = pp_start[pp]; o < pp_start[pp + 1]; ++o) {
          const int g = pose_gslot[obs_pose[o]];
          if (g >= 0 && std::find(seg.begin(), seg.end(), g) == seg.end()) seg.push_back(g);
        }
        for (size_t i = 0; i < seg.size(); ++i)
          for (size_t j = i + 1; j < seg.size(); ++j) graph.connect(seg[i], seg[j]);
      }
    }
    std::vector<int> order(ncam);  // order[k] = old slot at position k
    std::iota(order.begin(), order.end(), 0);
    if (pose_order) {
      std::vector<char> seen(probs[0].n_poses, 0);
      int k = 0;
      for (int i = 0; i < probs[0].n_poses; ++i) {
        const int c = pose_order[i];
        if (c < 0 || c >= probs[0].n_poses || seen[c])
          return fail(HIPBA_ERR_INVALID_ARGUMENT, "pose_order is not a permutation of the window's poses (entry %d = %d)", i, c);
        seen[c] = 1;
        if (pose_gslot[c] >= 0) order[k++] = pose_gslot[c];
      }
      if (k != ncam) return fail(HIPBA_ERR_INVALID_ARGUMENT, "pose_order lists %d of the %d variable poses", k, ncam);
    } else if (!all_poses_active) {
      // natural order or reverse Cuthill-McKee, whichever leaves the factorisation less to do
      double cost_id = 0.0, cost_rcm = 0.0;
      tile_envelope(graph, order, kCholNB, n_tiles, &cost_id);
      std::vector<int> rcm = reverse_cuthill_mckee(graph);
      tile_envelope(graph, rcm, kCholNB, n_tiles, &cost_rcm);
      if (cost_rcm < 0.9 * cost_id) order.swap(rcm);
    }
    R->env_first = tile_envelope(graph, order, kCholNB, n_tiles);
    std::vector<int> new_slot(ncam);
    for (int k = 0; k < ncam; ++k) new_slot[order[k]] = k;
    for (long long c = 0; c < NP; ++c)
      if (pose_gslot[c] >= 0) pose_gslot[c] = new_slot[pose_gslot[c]];
    for (long long c = 0; c < NP; ++c)
      if (pose_gslot[c] >= 0) slot_pose[pose_gslot[c]] = (int)c;
  }
  // sort each landmark's segment by slot (constant poses, slot -1, first; ties by pose index): the pair lists want
  // slot_i <= slot_j for i < j, and repeated observations by one pose next to each other
  for (int pp = 0; pp < n_pm_points && (dense_path || seg_unsorted); ++pp) {
    const long long s0 = pp_start[pp], s1 = pp_start[pp + 1];
    for (long long i = s0 + 1; i < s1; ++i) {
      const int cp = obs_pose[i];
      const long long key = ((long long)pose_gslot[cp] << 32) + cp;
      const double u = obs_uv[2 * i], v = obs_uv[2 * i + 1];
      long long j = i - 1;
      while (j >= s0 && ((long long)pose_gslot[obs_pose[j]] << 32) + obs_pose[j] > key) {
        obs_pose[j + 1] = obs_pose[j];
        obs_uv[2 * (j + 1)] = obs_uv[2 * j]; obs_uv[2 * (j + 1) + 1] = obs_uv[2 * j + 1];
        --j;
      }
      obs_pose[j + 1] = cp; obs_uv[2 * (j + 1)] = u; obs_uv[2 * (j + 1) + 1] = v;
    }
  }

  SETUP_MARK("pose order");
  // ---- blocks + pair lists -----------------------------------------------------------------------
  std::vector<BlockDesc> blocks;
  std::vector<unsigned short> pairs;
  // The pair list of a block: (i, j), i <= j, over the observations of each of its landmarks whose poses are
  // variable.  Only the sizes are fixed here (every list 16-byte aligned: blocks load them as uint4); the lists are
  // written further down, already in the order the accumulation mode wants.
  long long pairs_total = 0, pairs_all = 0;
  bool has_dup = false;  // some variable pose observes some landmark more than once
  std::vector<LongDesc> longs;
  for (int w = 0; w < nw; ++w) {
    wd[w].pm_blk_begin = (int)blocks.size();
    int pp = wd[w].pp_begin;
    while (pp < wd[w].pp_end) {
      BlockDesc d{};
      d.obs_begin = pp_start[pp];
      pairs_total = (pairs_total + 7) & ~7ll;
      d.pair_begin = pairs_total;
      d.window = w;
      d.kind = 0;
      int n = 0, np = 0;
      if (pp_start[pp + 1] - pp_start[pp] > kBlock) {
        // a landmark with more observations than a block holds: a run of blocks of its own, flagged with the index
        // (+1) of its LongDesc; the per-landmark kernels leave those blocks to k_long_point
        LongDesc L{};
        L.obs_begin = pp_start[pp];
        L.n_obs = (int)(pp_start[pp + 1] - pp_start[pp]);
        L.pp = pp;
        L.blk_begin = (int)blocks.size();
        for (int left = L.n_obs; left > 0; left -= kBlock) {
          BlockDesc e = d;
          e.obs_begin = L.obs_begin + (L.n_obs - left);
          e.n_obs = std::min(left, kBlock);
          e.n_pairs = 0;
          e.pad = (int)longs.size() + 1;
          blocks.push_back(e);
        }
        L.n_blk = (int)blocks.size() - L.blk_begin;
        longs.push_back(L);
        const long long v = nvar[pp];
        pairs_all += v * (v + 1) / 2;
        ++pp;
        continue;
      }
      while (pp < wd[w].pp_end && n + (int)(pp_start[pp + 1] - pp_start[pp]) <= kBlock) {
        const int s1 = n + (int)(pp_start[pp + 1] - pp_start[pp]);
        const int v = nvar[pp];
        np += v * (v + 1) / 2;
        // a pose that observes the landmark m times (m residual blocks, legal in ceres::Problem) adds m (m - 1) / 2
        // mirrored pairs: see generate_pairs.  The segment is sorted by pose, so repeats are adjacent.
        for (int i = n, run = 1; i < s1 && !segments_known_clean; ++i) {
          const bool same = i + 1 < s1 && obs_pose[d.obs_begin + i + 1] == obs_pose[d.obs_begin + i];
          if (same) { ++run; continue; }
          if (pose_gslot[obs_pose[d.obs_begin + i]] >= 0) { np += run * (run - 1) / 2; has_dup |= run > 1; }
          run = 1;
        }
        n = s1;
        ++pp;
      }
      d.n_obs = n;
      d.n_pairs = np;
      pairs_total += np;
      pairs_all += np;
      blocks.push_back(d);
    }
    wd[w].pm_blk_end = (int)blocks.size();
  }
  // Small windows (the reduced system solved in LDS) of at most 32 poses, no pose observing a landmark twice: every
  // lane of the elimination owns one block of S and finds its products itself from the landmarks' slot masks
  // (point_schur_body, MODE 4) -- no pair list, no run list, nothing to group here or to upload.
  // HIPBA_PAIR_LISTS=1 keeps the lists (A/B timing).
  static const bool keep_pair_lists = getenv("HIPBA_PAIR_LISTS") != nullptr;
  // (a single window with millions of products goes to the gathered elimination below, as before)
  const bool gathered_instead = nw == 1 && max_ncam > 20 && pairs_all + 8 >= 2000000;
  const bool dest_owned = !keep_pair_lists && !has_dup && !has_long && !gathered_instead && max_ncam >= 1 && max_ncam <= 32 &&
                          6 * max_ncam <= kSmallMaxN;
  pairs_total = ((pairs_total + 7) & ~7ll) + 8;              // whole-uint4 reads of the last list stay inside
  pairs.assign(dest_owned ? 8 : (size_t)pairs_total, kPairPad);
  // plain generation order (landmark by landmark, i then j): the gathered mode and the sorting fallback start from it.
  // A pair (i, j), i < j, of two DIFFERENT poses goes to an off-diagonal block of the reduced matrix, of which only
  // the upper one is kept; two observations of the SAME pose both go to its diagonal block, which needs
  // Y_i W_j^T + Y_j W_i^T: such a pair is listed in both orders.
  auto generate_pairs = [&](const BlockDesc& d) {
    const int w = d.window;
    (void)w;
    long long q = d.pair_begin;
    // landmark boundaries inside the block: consecutive observations of one landmark share obs_dp
    for (int s0 = 0; s0 < d.n_obs;) {
      int s1 = s0 + 1;
      while (s1 < d.n_obs && obs_dp[d.obs_begin + s1] == obs_dp[d.obs_begin + s0]) ++s1;
      for (int i = s0; i < s1; ++i) {
        if (pose_gslot[obs_pose[d.obs_begin + i]] < 0) continue;
        for (int j = i; j < s1; ++j) {
          if (pose_gslot[obs_pose[d.obs_begin + j]] < 0) continue;
          pairs[q++] = (unsigned short)(i | (j << 8));
          if (j > i && obs_pose[d.obs_begin + j] == obs_pose[d.obs_begin + i]) pairs[q++] = (unsigned short)(j | (i << 8));
        }
      }
      s0 = s1;
    }
  };
  const int n_blk_pm = (int)blocks.size();
  {
    long long cur = n_pm_obs;
    for (int w = 0; w < nw; ++w) {
      wd[w].vc_blk_begin = (int)blocks.size();
      for (int left = vc_window_count[w]; left > 0;) {
        BlockDesc d{};
        d.obs_begin = cur; d.n_obs = std::min(left, kBlock); d.window = w; d.kind = 1;
        blocks.push_back(d);
        cur += d.n_obs; left -= d.n_obs;
      }
      wd[w].vc_blk_end = (int)blocks.size();
    }
  }
  const int n_blk_vc = (int)blocks.size() - n_blk_pm;
  {
    long long cur = n_act_obs;
    for (int w = 0; w < nw; ++w) {
      wd[w].cc_blk_begin = (int)blocks.size();
      for (int left = cc_window_count[w]; left > 0;) {
        BlockDesc d{};
        d.obs_begin = cur; d.n_obs = std::min(left, kBlock); d.window = w; d.kind = 2;
        blocks.push_back(d);
        cur += d.n_obs; left -= d.n_obs;
      }
      wd[w].cc_blk_end = (int)blocks.size();
    }
  }
  const int n_blk_cc = (int)blocks.size() - n_blk_pm - n_blk_vc;
  const int n_blocks = (int)blocks.size();

  SETUP_MARK("blocks+pairs");
  // ---- gathered accumulation (one large window): observation pairs listed per destination block ---------
  // fp64 atomics into S run at ~1 TB/s of added bytes; beyond a few million 6x6 products it is faster to let
  // every block of S sum its own products, each recomputed from the landmark's record and the two poses.
  std::vector<uint2> pair_obs, pair_pt;
  std::vector<long long> chunk_start;
  std::vector<uint4> lane_jobs;
  const bool can_stage = nw == 1 && max_ncam > 20 && max_ncam <= 0xffff && n_pm_obs < 0xffffffffll;
  if (has_long && !can_stage)
    return fail(HIPBA_ERR_UNSUPPORTED, "a landmark with more than %d observations needs a single window of more than 20 "
                "variable poses (the gathered elimination); this call has %d window(s), at most %d variable poses", kBlock,
                nw, max_ncam);
  const bool staged = can_stage && (has_long || pairs_all + 8 >= 2000000);
  if (staged) {
    // every pair (i, j), i <= j, of the observations of a landmark by variable poses, straight from the landmark's
    // segment (blocks play no part here, so a landmark may be as long as it likes); two observations by the SAME
    // pose go to a diagonal block, which needs the product in both orders
    const int ncam = wd[0].ncam;
    const int cam_base = wd[0].cam_base;
    const long long n_dest = (long long)ncam * (ncam + 1) / 2;
    std::vector<long long> dest_start(n_dest + 1, 0);
    auto for_each_pair = [&](auto&& f) {
      for (int pp = 0; pp < n_pm_points; ++pp) {
        for (long long oi = pp_start[pp]; oi < pp_start[pp + 1]; ++oi) {
          const int si = pose_gslot[obs_pose[oi]] - cam_base;
          if (si < 0) continue;
          const long long row = tri_index(si, si, ncam) - si;
          for (long long oj = oi; oj < pp_start[pp + 1]; ++oj) {
            const int sj = pose_gslot[obs_pose[oj]] - cam_base;
            if (sj < 0) continue;
            f(pp, oi, oj, si, sj, row + sj);
            if (oj > oi && sj == si) f(pp, oj, oi, si, sj, row + sj);
          }
        }
      }
    };
    for_each_pair([&](int, long long, long long, int, int, long long dst) { dest_start[dst + 1]++; });
    {
      // the gather stores every off-diagonal block that has a pair and the initialisation every diagonal one
      bool all = true;
      for (long long i = 0; i < n_dest && all; ++i) all = dest_start[i + 1] > 0;
      R->S_fully_written = all;
    }
    for (long long i = 0; i < n_dest; ++i) dest_start[i + 1] += dest_start[i];
    std::vector<long long> cur(dest_start.begin(), dest_start.end() - 1);
    pair_obs.resize((size_t)dest_start[n_dest]);
    pair_pt.resize(pair_obs.size());
    for_each_pair([&](int pp, long long oi, long long oj, int si, int sj, long long dst) {
      const long long at = cur[dst]++;
      pair_obs[at] = make_uint2((unsigned)oi, (unsigned)oj);
      // bit 31 of the landmark word: the pair of an observation with itself, which also carries the observation's
      // term of the reduced right-hand side (k_schur_gather)
      pair_pt[at] = make_uint2((unsigned)pp | (oi == oj ? 0x80000000u : 0u), ((unsigned)si << 16) | (unsigned)sj);
    });
    // one wave per chunk of about kGatherChunk pairs; a chunk holds whole destinations, so every block of S
    // has exactly one writer
    // (begin, end) per chunk, longest first: a diagonal block alone has as many pairs as its camera has
    // observations, and such a chunk started last would be the tail of the launch
    const long long kGatherChunk = 1024;
    std::vector<std::pair<long long, long long>> chunks;
    // Off-diagonal blocks have a few dozen pairs each (two poses share the landmarks both were drawn for): one LANE
    // sums such a block (k_schur_gather_lanes), longest blocks first so that the 64 lanes of a wave finish together.
    // Diagonal blocks -- as many pairs as the pose has observations, and the right-hand side -- and off-diagonal
    // ones too long for a lane keep the wave-per-chunk kernel.  HIPBA_GATHER_WAVES=1: everything on that kernel.
    static const bool waves_only = getenv("HIPBA_GATHER_WAVES") != nullptr;
    const long long kLaneMaxPairs = 512;
    const bool lanes = !waves_only && dest_start[n_dest] < 0xffffffffll;
    std::vector<char> by_lane(lanes ? (size_t)n_dest : 0, 0);
    if (lanes) {
      long long dd = 0;
      for (int si = 0; si < ncam; ++si) {
        for (int sj = si; sj < ncam; ++sj, ++dd) {
          const long long len = dest_start[dd + 1] - dest_start[dd];
          if (sj == si || len == 0 || len > kLaneMaxPairs) continue;
          by_lane[dd] = 1;
          lane_jobs.push_back(make_uint4((unsigned)dest_start[dd], (unsigned)len, ((unsigned)si << 16) | (unsigned)sj, 0u));
        }
      }
      std::stable_sort(lane_jobs.begin(), lane_jobs.end(), [](const uint4& x, const uint4& y) { return x.y > y.y; });
    }
    long long cbeg = 0;
    for (long long dd = 0; dd < n_dest; ++dd) {
      if (lanes && by_lane[dd]) {
        // (a chunk is a contiguous run of whole blocks: it ends in front of a block that a lane takes)
        if (cbeg != dest_start[dd]) chunks.emplace_back(cbeg, dest_start[dd]);
        cbeg = dest_start[dd + 1];
      } else if (dest_start[dd + 1] - cbeg >= kGatherChunk) {
        chunks.emplace_back(cbeg, dest_start[dd + 1]);
        cbeg = dest_start[dd + 1];
      }
    }
    if (cbeg != dest_start[n_dest]) chunks.emplace_back(cbeg, dest_start[n_dest]);
    std::stable_sort(chunks.begin(), chunks.end(), [](const std::pair<long long, long long>& x, const std::pair<long long, long long>& y) {
      return x.second - x.first > y.second - y.first;
    });
    for (const auto& c : chunks) { chunk_start.push_back(c.first); chunk_start.push_back(c.second); }
    pairs.assign(8, kPairPad);  // the blocks themselves need no pair lists in this mode
  }

  SETUP_MARK("gathered lists");
  // the camera-major lists: built meanwhile by a worker (small windows), or here now that the slots are final
  if (!side_lists_done) build_side_lists();

  // ---- atomic / LDS accumulation: each block's pair list grouped by destination block and cut into runs
  // of at most kRunMax pairs with one destination; a lane group sums a run in registers and adds it to
  // the reduced matrix once ------------------------------------------------------------------------
  std::vector<RunDesc> runs;
  const long long n_dest_max = (long long)max_ncam * (max_ncam + 1) / 2;
  if (!staged && !dest_owned && n_blk_pm > 0) {
    // a keyframe window has ~10^5 pairs: worth four short-lived threads (grouping them is the longest host step
    // of hipba_solve); large problems use up to sixteen
    const int n_thr = (int)std::min<size_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())),
                                            pairs.size() < 1000000 ? std::min<size_t>(8, 1 + pairs.size() / 12000)
                                                                   : 1 + pairs.size() / 1000000);
    std::vector<std::vector<RunDesc>> runs_of(n_thr);
    std::vector<std::vector<int>> nruns_of(n_thr);
    auto group_range = [&](int ti, int b0, int b1) {
      // counting sort over the destinations a block touches, in order of first appearance
      std::vector<int> count, touched;
      std::vector<std::pair<long long, int>> keyed;   // fallback when a count array would be too large
      const bool counting = n_dest_max <= (1ll << 22);
      if (counting) count.assign(n_dest_max, 0);
      runs_of[ti].reserve(pairs.size() / (n_thr * 2) + 16);
      nruns_of[ti].reserve(b1 - b0);
      for (int bi = b0; bi < b1; ++bi) {
        const BlockDesc& d = blocks[bi];
        const int ncam = wd[d.window].ncam, cam_base = wd[d.window].cam_base;
        int nr = 0;
        auto emit = [&](long long dest, int first, int n) {
          for (int o = 0; o < n; o += kRunMax) {
            runs_of[ti].push_back(RunDesc{((unsigned)(first + o) << 8) | (unsigned)std::min(kRunMax, n - o), (unsigned)dest});
            ++nr;
          }
        };
        int lane_slot[kBlock];  // window slot of each observation of the block (looked up once, not per pair); -1: constant pose
        for (int i = 0; i < d.n_obs; ++i) {
          const int g = pose_gslot[obs_pose[d.obs_begin + i]];
          lane_slot[i] = g < 0 ? -1 : g - cam_base;
        }
        if (counting) {
          // Two sweeps over the block's pairs in generation order (landmark by landmark, i then j >= i), neither of
          // which stores a key: the first counts the pairs of every destination and notes the destinations in order of
          // first appearance, the second drops each pair at its destination's cursor.
          long long row_base[kBlock];  // tri_index(slot_i, slot_j) = row_base[i] + slot_j
          int lm_end[kBlock];          // one past the last observation of lane i's landmark
          for (int s0 = 0; s0 < d.n_obs;) {
            int s1 = s0 + 1;
            while (s1 < d.n_obs && obs_dp[d.obs_begin + s1] == obs_dp[d.obs_begin + s0]) ++s1;
            for (int i = s0; i < s1; ++i) {
              lm_end[i] = s1;
              row_base[i] = lane_slot[i] < 0 ? 0 : tri_index(lane_slot[i], lane_slot[i], ncam) - lane_slot[i];
            }
            s0 = s1;
          }
          touched.clear();
          for (int i = 0; i < d.n_obs; ++i) {
            if (lane_slot[i] < 0) continue;
            for (int j = i; j < lm_end[i]; ++j) {
              if (lane_slot[j] < 0) continue;
              const int kq = (int)(row_base[i] + lane_slot[j]);
              if (count[kq]++ == 0) touched.push_back(kq);
              if (j > i && lane_slot[j] == lane_slot[i]) count[kq]++;  // same pose twice: both orders (generate_pairs)
            }
          }
          int run = 0;
          for (int kq : touched) {
            const int c = count[kq];
            emit(kq, run, c);
            count[kq] = run;
            run += c;
          }
          unsigned short* out = pairs.data() + d.pair_begin;
          for (int i = 0; i < d.n_obs; ++i) {
            if (lane_slot[i] < 0) continue;
            for (int j = i; j < lm_end[i]; ++j) {
              if (lane_slot[j] < 0) continue;
              int& at = count[(int)(row_base[i] + lane_slot[j])];
              out[at++] = (unsigned short)(i | (j << 8));
              if (j > i && lane_slot[j] == lane_slot[i]) out[at++] = (unsigned short)(j | (i << 8));
            }
          }
          for (int kq : touched) count[kq] = 0;
        } else {
          generate_pairs(d);
          auto dest_of_pair = [&](unsigned short pr) { return tri_index(lane_slot[pr & 0xff], lane_slot[pr >> 8], ncam); };
          keyed.resize(d.n_pairs);
          for (int q = 0; q < d.n_pairs; ++q) keyed[q] = {dest_of_pair(pairs[d.pair_begin + q]), pairs[d.pair_begin + q]};
          std::sort(keyed.begin(), keyed.end());
          for (int q = 0, q0 = 0; q <= d.n_pairs; ++q) {
            if (q < d.n_pairs) pairs[d.pair_begin + q] = (unsigned short)keyed[q].second;
            if (q == d.n_pairs || keyed[q].first != keyed[q0].first) {
              if (q > q0) emit(keyed[q0].first, q0, q - q0);
              q0 = q;
            }
          }
        }
        nruns_of[ti].push_back(nr);
      }
    };
    auto lo = [&](int i) { return (int)((long long)n_blk_pm * i / n_thr); };
    if (n_thr <= 1) {
      group_range(0, 0, n_blk_pm);
    } else if (n_thr <= 8) {
      if (worker_busy) { (void)h->workers.wait(); worker_busy = false; }
      // the handle's sleeping workers (and this thread) take the blocks
      if (!h->workers.run(n_thr, [&](int i) { group_range(i, lo(i), lo(i + 1)); }))
        return fail(HIPBA_ERR_OUT_OF_MEMORY, "host analysis of the pair lists ran out of memory");
    } else {
      // short-lived threads for a large problem; an exception of a part stays inside its thread, and a thread that
      // cannot be started ends the analysis after the ones already running have been joined
      std::atomic<bool> failed{false};
      std::vector<std::thread> pool;
      pool.reserve(n_thr);
      for (int i = 0; i < n_thr && !failed; ++i) {
        try {
          pool.emplace_back([&, i] {
            try { group_range(i, lo(i), lo(i + 1)); } catch (...) { failed = true; }
          });
        } catch (...) {
          failed = true;
        }
      }
      for (auto& th : pool) th.join();
      if (failed) return fail(HIPBA_ERR_OUT_OF_MEMORY, "host analysis of the pair lists ran out of memory or threads");
    }
    // splice the per-thread run lists, every block's list starting on a 16-byte boundary
    for (int i = 0; i < n_thr; ++i) {
      size_t src = 0;
      for (int bi = lo(i); bi < lo(i + 1); ++bi) {
        const int nr = nruns_of[i][bi - lo(i)];
        if (runs.size() % 2) runs.push_back(RunDesc{0, 0});
        blocks[bi].run_begin = (long long)runs.size();
        blocks[bi].n_runs = nr;
        runs.insert(runs.end(), runs_of[i].begin() + src, runs_of[i].begin() + src + nr);
        src += nr;
      }
    }
    runs.push_back(RunDesc{0, 0});
    runs.push_back(RunDesc{0, 0});
  }

  SETUP_MARK("runs");
  for (int w = 0; w < nw; ++w) {
    long long na = 0;
    for (int b = wd[w].pm_blk_begin; b < wd[w].pm_blk_end; ++b) na += blocks[b].n_obs;
    for (int b = wd[w].vc_blk_begin; b < wd[w].vc_blk_end; ++b) na += blocks[b].n_obs;
    R->meta[w].n_act_obs = na;
    R->meta[w].dp_of.assign(dp_of.begin() + pt_base[w], dp_of.begin() + pt_base[w + 1]);
  }

  SETUP_MARK("camera-major");
  SETUP_MARK("parameters");
  // ---- plan the arena ----------------------------------------------------------------------------
  P.n_poses = (int)NP; P.n_points = (int)NPT; P.n_act_poses = n_slots; P.n_pm_points = n_pm_points;
  P.n_pm_obs = n_pm_obs; P.n_act_obs = n_act_obs; P.n_all_obs = n_all;
  P.n_blk_pm = n_blk_pm; P.n_blk_vc = n_blk_vc; P.n_blk_cc = n_blk_cc;
  P.log_cap = log_cap;
  R->max_ncam = max_ncam; R->n_blocks = n_blocks;
  R->S_doubles = S_doubles;
  for (int w = 0; w < nw; ++w) R->max_pm_blocks = std::max(R->max_pm_blocks, wd[w].pm_blk_end - wd[w].pm_blk_begin);

  ArenaPlan up;  // uploaded region (one host->device copy)
  int *d_pose_window, *d_pose_gslot, *d_slot_pose, *d_cm_dp, *d_obs_pose, *d_obs_dp;
  long long *d_cm_start, *d_pp_start;
  double *d_cm_uv, *d_obs_uv;
  BlockDesc* d_blocks; unsigned short* d_pairs; WinDesc* d_wd; RunDesc* d_runs;
  uint2 *d_pair_obs = nullptr, *d_pair_pt = nullptr; long long* d_chunk_start = nullptr; uint4* d_lane_jobs = nullptr;
  // parameters (both buffers) and the window states first and next to each other: they arrive with the upload (the
  // candidate buffers as copies, the states as zeros) and leave in one copy at the end of a solve
  up.add(&P.poses[0], poses.size());
  up.add(&P.points[0], points.size());
  up.add(&P.poses[1], poses.size());
  up.add(&P.points[1], points.size());
  up.add(&P.ws, (size_t)nw);
  R->tail_span = up.total;
  up.add(&d_pose_window, std::max<size_t>(pose_window.size(), 1));
  up.add(&d_pose_gslot, std::max<size_t>(pose_gslot.size(), 1));
  up.add(&d_slot_pose, std::max<size_t>(slot_pose.size(), 1));
  up.add(&d_cm_start, cm_start.size());
  up.add(&d_cm_dp, cm_dp.size());
  up.add(&d_cm_uv, cm_uv.size());
  up.add(&d_obs_pose, obs_pose.size());
  up.add(&d_obs_dp, obs_dp.size());
  up.add(&d_obs_uv, obs_uv.size());
  up.add(&d_pp_start, pp_start.size());
  up.add(&d_blocks, std::max<size_t>(blocks.size(), 1));
  up.add(&d_pairs, std::max<size_t>(pairs.size(), 1));
  up.add(&d_wd, wd.size());
  up.add(&d_runs, std::max<size_t>(runs.size(), 1));
  if (staged) { up.add(&d_pair_obs, pair_obs.size()); up.add(&d_pair_pt, pair_pt.size()); up.add(&d_chunk_start, chunk_start.size()); up.add(&d_lane_jobs, lane_jobs.size()); }
  LongDesc* d_longs = nullptr;
  up.add(&d_longs, std::max<size_t>(longs.size(), 1));
  const size_t upload_bytes = up.total;
  R->upload_bytes = (long long)upload_bytes;
  ArenaPlan work = up;  // continues after the uploaded region
  work.items.clear();
  const size_t ns = std::max(n_slots, 1), npp = std::max(n_pm_points, 1);
  work.add(&R->poses0, poses.size());
  work.add(&R->points0, points.size());
  work.add(&P.U, 36 * ns);
  work.add(&P.gc, 6 * ns);
  work.add(&P.scale_c, 6 * ns);
  work.add(&P.b, 6 * ns);
  work.add(&P.yc, 6 * ns);
  work.add(&P.Vinv, 6 * npp);
  work.add(&P.gp, 3 * npp);
  work.add(&P.scale_p, 3 * npp);
  work.add(&P.S, (size_t)std::max<long long>(S_doubles, 1));
  work.add(&P.partial, (size_t)2 * kPartialStride * std::max(n_blocks, 1));
  work.add(&P.gpart, (size_t)2 * std::max(n_blk_pm, 1));
  work.add(&P.wsum, (size_t)8 * nw);
  if (nw == 1 && 6 * max_ncam <= kSmallMaxN) work.add(&P.tile_src, (size_t)(kSmallMaxN / 16) * (kSmallMaxN / 16 + 1) / 2 * 256);
  work.add(&P.log, (size_t)std::max(log_cap, 1) * nw);
  if (staged) { work.add(&P.prec, 16 * npp); work.add(&P.fw, (size_t)n_pm_obs); work.add(&P.cam_tab, 18 * ns); }
  char* big_base_dummy = nullptr;
  size_t big_bytes = 0;
  if (6 * max_ncam > kSmallMaxN) {
    big_bytes = big_work_bytes(6 * max_ncam, max_ncam);
    work.add(&big_base_dummy, big_bytes);
    R->has_big = true;
  }
  const size_t total = work.total;

  SETUP_MARK("plan");
  // ---- allocate + upload -------------------------------------------------------------------------
  int rc = HIPBA_OK;
  if (use_handle_arena) {
    rc = ensure_arena(h, total);
    if (rc) return rc;
    R->arena = h->arena; R->owns_arena = false;
  } else {
    hipError_t e = hipMalloc(&R->arena, total);
    if (e != hipSuccess) { R->arena = nullptr; return fail(HIPBA_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes): %s", total, hipGetErrorString(e)); }
    R->owns_arena = true;
  }
  R->arena_bytes = total;
  rc = ensure_staging(h, upload_bytes);
  if (rc) return rc;
  up.bind(R->arena);
  work.bind(R->arena);
  if (R->has_big) big_work_bind(R->big, 6 * max_ncam, max_ncam, big_base_dummy);
  P.pose_window = d_pose_window; P.pose_gslot = d_pose_gslot; P.slot_pose = d_slot_pose;
  P.cm_start = d_cm_start; P.cm_dp = d_cm_dp; P.cm_uv = d_cm_uv;
  P.obs_pose = d_obs_pose; P.obs_dp = d_obs_dp; P.obs_uv = d_obs_uv; P.pp_obs_start = d_pp_start;
  P.blocks = d_blocks; P.pairs = d_pairs; P.wd = d_wd; P.runs = d_runs;
  P.pair_obs = d_pair_obs; P.pair_pt = d_pair_pt; P.chunk_start = d_chunk_start;
  P.n_chunks = staged ? (int)(chunk_start.size() / 2) : 0;
  P.lane_jobs = d_lane_jobs; P.n_lane_jobs = staged ? (int)lane_jobs.size() : 0;
  P.dest_owned = dest_owned ? 1 : 0;
  if (!staged) { P.prec = nullptr; P.fw = nullptr; P.cam_tab = nullptr; }

  char* st = static_cast<char*>(h->staging);
  auto stage = [&](const void* dev_ptr, const void* src, size_t bytes) {
    if (bytes) memcpy(st + (static_cast<const char*>(dev_ptr) - static_cast<const char*>(R->arena)), src, bytes);
  };
  stage(P.poses[0], poses.data(), poses.size() * 8);
  stage(P.points[0], points.data(), points.size() * 8);
  stage(P.poses[1], poses.data(), poses.size() * 8);
  stage(P.points[1], points.data(), points.size() * 8);
  memset(st + (reinterpret_cast<const char*>(P.ws) - static_cast<const char*>(R->arena)), 0, sizeof(WinState) * nw);
  stage(d_pose_window, pose_window.data(), pose_window.size() * 4);
  stage(d_pose_gslot, pose_gslot.data(), pose_gslot.size() * 4);
  stage(d_slot_pose, slot_pose.data(), slot_pose.size() * 4);
  stage(d_cm_start, cm_start.data(), cm_start.size() * 8);
  stage(d_cm_dp, cm_dp.data(), cm_dp.size() * 4);
  stage(d_cm_uv, cm_uv.data(), cm_uv.size() * 8);
  stage(d_obs_pose, obs_pose.data(), obs_pose.size() * 4);
  stage(d_obs_dp, obs_dp.data(), obs_dp.size() * 4);
  stage(d_obs_uv, obs_uv.data(), obs_uv.size() * 8);
  stage(d_pp_start, pp_start.data(), pp_start.size() * 8);
  stage(d_blocks, blocks.data(), blocks.size() * sizeof(BlockDesc));
  stage(d_pairs, pairs.data(), pairs.size() * 2);
  stage(d_wd, wd.data(), wd.size() * sizeof(WinDesc));
  stage(d_runs, runs.data(), runs.size() * sizeof(RunDesc));
  stage(d_longs, longs.data(), longs.size() * sizeof(LongDesc));
  P.longs = d_longs; P.n_long = (int)longs.size();
  if (staged) {
    stage(d_pair_obs, pair_obs.data(), pair_obs.size() * sizeof(uint2));
    stage(d_pair_pt, pair_pt.data(), pair_pt.size() * sizeof(uint2));
    stage(d_chunk_start, chunk_start.data(), chunk_start.size() * 8);
    stage(d_lane_jobs, lane_jobs.data(), lane_jobs.size() * sizeof(uint4));
  }
  hipStream_t s = h->stream;
  SETUP_MARK("fill staging");
  hipError_t e = hipMemcpyAsync(R->arena, h->staging, upload_bytes, hipMemcpyHostToDevice, s);
  // development switch: HIPBA_POISON_WORKSPACE=1 fills every work array behind the uploaded part with NaN patterns, so
  // that a kernel reading a word nobody has written shows up in the result (a reused arena holds the last problem's)
  static const bool poison = getenv("HIPBA_POISON_WORKSPACE") != nullptr;
  if (poison && e == hipSuccess && total > upload_bytes)
    e = hipMemsetAsync(static_cast<char*>(R->arena) + upload_bytes, 0xFF, total - upload_bytes, s);
  if (!use_handle_arena) {  // (a one-shot solve is never reset)
    if (e == hipSuccess) e = hipMemcpyAsync(R->poses0, P.poses[0], poses.size() * 8, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(R->points0, P.points[0], points.size() * 8, hipMemcpyDeviceToDevice, s);
  }
  if (e == hipSuccess && R->has_big && !R->env_first.empty()) {
    R->env_first.push_back(0);  // the right-hand side's tile row is dense
    e = hipMemcpyAsync(R->big.first, R->env_first.data(), sizeof(int) * R->env_first.size(), hipMemcpyHostToDevice, s);
  }
  // The staging buffer is reused by the next call: a resident upload waits for the copy here; a one-shot solve goes
  // straight on to queue its kernels behind it and waits once, at its end (hipba_solve_batched, also when it fails).
  if (e == hipSuccess && !use_handle_arena) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(HIPBA_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
  SETUP_MARK("queue copies");
  *out = owner.release();
  return HIPBA_OK;
}

void print_progress(const hipba_iteration* log, int n) {
  printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter\n");
  for (int i = 0; i < n; ++i) {
    const hipba_iteration& it = log[i];
    printf("%4d % .6e  % .2e   %.2e   %.2e  % .2e   %.2e   %4d%s\n", it.iteration, it.cost, it.cost_change,
           it.gradient_max_norm, it.step_norm, it.relative_decrease, it.trust_region_radius,
           it.linear_solver_iterations, it.step_is_successful ? "" : (it.step_is_valid ? "  (rejected)" : "  (invalid)"));
  }
}

// ---- the LM loop ---------------------------------------------------------------------------------
int run_lm(hipba_handle_s* h, hipba_device_problem_s* R, const hipba_options& o, const hipba_comm* comm,
           hipba_summary* summaries, bool fetch_tail = false) {
  const double t0 = now_s();
  DevP P = R->P;
  hipStream_t s = h->stream;
  const LmParams lm = make_lm(o);
  // (HIPBA_SHARD_SINGLE_RANK: a communicator of one rank takes the sharded path too -- every all-reduce is then the
  // identity -- so that the whole exchange, the second stream included, can be rehearsed on one GPU)
  const bool sharded = comm && (comm->world_size > 1 || (comm->world_size == 1 && getenv("HIPBA_SHARD_SINGLE_RANK")));
  if (sharded && P.n_windows != 1) return fail(HIPBA_ERR_INVALID_ARGUMENT, "sharded solves take exactly one window");
  if (sharded && !comm->allreduce) return fail(HIPBA_ERR_INVALID_ARGUMENT, "comm->allreduce is null");
  P.sharded = sharded ? 1 : 0;
  const int owner = (!sharded || comm->rank == 0) ? 1 : 0;
  const int n_act_blk = P.n_blk_pm + P.n_blk_vc;
  const int max_n = 6 * R->max_ncam;
  const bool small = max_n <= kSmallMaxN;
  int solver = o.linear_solver;
  if (small) solver = HIPBA_LINEAR_DENSE_CHOLESKY;
  else if (solver == HIPBA_LINEAR_AUTO) solver = HIPBA_LINEAR_DENSE_CHOLESKY;
  // windows the in-LDS solver takes also accumulate b (and a small enough S) in LDS
  const int small_ncam = small ? R->max_ncam : 0;
  const int schur_parts = schur_parts_for(P.n_windows, R->max_pm_blocks);
  // gathered elimination (one writer per block of the reduced matrix) + the dense factorisation: the blocks go
  // straight into the factorisation's matrix, no packed copy
  bool direct = P.n_chunks > 0 && !small && solver == HIPBA_LINEAR_DENSE_CHOLESKY;

  auto allreduce = [&](double* buf, size_t n, int op) -> int {
    if (!sharded) return 0;
    return comm->allreduce(comm->ctx, buf, n, op, (void*)s);
  };
#define AR(buf, n, op)                                                                  \
  do {                                                                                  \
    if (allreduce((buf), (n), (op))) return fail(HIPBA_ERR_COMM, "all-reduce callback failed"); \
  } while (0)

  // The dense factorisation skips what lies outside the envelope of the reduced matrix.  A shard knows the couplings
  // of its own landmarks only: the ranks merge their envelopes (element-wise minimum of the first tile columns)
  // before the first factorisation.
  bool env_ready = R->has_big && !R->env_first.empty();
  // Sharded + gathered elimination + dense factorisation: every rank writes its blocks straight into the dense
  // matrix and the ranks exchange only the 128x128 tiles inside the (merged) envelope, packed into the buffer of the
  // packed reduced matrix.  All ranks must take the same path, so whether every rank can rides along in the
  // envelope's all-reduce as one more word (a shard small enough for the atomic elimination, or an envelope too
  // wide for the buffer, sends everybody down the packed path).
  int n_env_tiles = 0;
  TileFeed feed;  // n_chunks > 0: the tiles travel while the factorisation runs
  if (sharded) direct = direct && env_ready;
  if (env_ready && sharded) {
    const size_t nt1 = R->env_first.size();
    std::vector<double> neg(nt1 + 1);
    for (size_t i = 0; i < nt1; ++i) neg[i] = -(double)R->env_first[i];
    neg[nt1] = direct ? -1.0 : 0.0;  // MAX over ranks: 0 as soon as one rank cannot
    HIPBA_HIP(hipMemcpyAsync(P.yc, neg.data(), sizeof(double) * (nt1 + 1), hipMemcpyHostToDevice, s));
    AR(P.yc, nt1 + 1, HIPBA_REDUCE_MAX);
    HIPBA_HIP(hipMemcpyAsync(neg.data(), P.yc, sizeof(double) * (nt1 + 1), hipMemcpyDeviceToHost, s));
    HIPBA_HIP(hipStreamSynchronize(s));
    for (size_t i = 0; i < nt1; ++i) R->env_first[i] = (int)(-neg[i]);
    HIPBA_HIP(hipMemcpyAsync(R->big.first, R->env_first.data(), sizeof(int) * nt1, hipMemcpyHostToDevice, s));
    direct = neg[nt1] < 0.0;
    if (direct) {
      // by tile column, the order the factorisation consumes them in (TileFeed)
      std::vector<int2> tiles;
      std::vector<int> col_begin;
      const int nt = (int)nt1 - 1;
      for (int j = 0; j < nt; ++j) {
        col_begin.push_back((int)tiles.size());
        for (int i = j; i < nt; ++i)
          if (R->env_first[i] <= j) tiles.push_back(make_int2(i, j));
      }
      col_begin.push_back((int)tiles.size());
      // (the merged envelope is the same on every rank, and so is this test)
      if ((long long)tiles.size() * kCholNB * kCholNB > R->S_doubles) {
        direct = false;
      } else {
        n_env_tiles = (int)tiles.size();
        HIPBA_HIP(hipMemcpyAsync(R->big.tiles, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice, s));
        HIPBA_HIP(hipStreamSynchronize(s));  // `tiles` goes out of scope
        // The exchange in chunks of whole tile columns, about equal in tiles, that the factorisation picks up as it
        // goes: the first chunk is small (the factorisation cannot start before it is there), HIPBA_EXCHANGE_CHUNKS=1
        // brings back the exchange in one piece ahead of the factorisation.
        const int want_chunks = getenv("HIPBA_EXCHANGE_CHUNKS") ? atoi(getenv("HIPBA_EXCHANGE_CHUNKS")) : 8;
        const int n_chunks = std::min(std::min(want_chunks, (int)kMaxFeedChunks), nt);
        if (n_chunks > 1) {
          feed.n_chunks = 0;
          feed.packed = P.S;
          int col = 0;
          for (int c = 0; c < n_chunks && col < nt; ++c) {
            // chunk 0: half a share; the others split what is left evenly
            const int left = n_env_tiles - col_begin[col];
            const int share = c == 0 ? std::max(1, n_env_tiles / (2 * n_chunks)) : std::max(1, left / (n_chunks - c));
            int end = col + 1;
            while (end < nt && col_begin[end] - col_begin[col] < share) ++end;
            if (c == n_chunks - 1) end = nt;
            feed.first_col[feed.n_chunks] = col;
            feed.tile_begin[feed.n_chunks] = col_begin[col];
            ++feed.n_chunks;
            col = end;
          }
          feed.tile_begin[feed.n_chunks] = n_env_tiles;
          if (!h->xchg_ready_all) {
            // (created piece by piece; a failure part-way leaves what exists to hipba_destroy and is retried here)
            if (!h->xchg_stream) HIPBA_HIP(hipStreamCreateWithFlags(&h->xchg_stream, hipStreamNonBlocking));
            if (!h->xchg_packed) HIPBA_HIP(hipEventCreateWithFlags(&h->xchg_packed, hipEventDisableTiming));
            for (auto& ev : h->xchg_ready)
              if (!ev) HIPBA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            h->xchg_ready_all = true;
          }
          for (int c = 0; c < feed.n_chunks; ++c) feed.ready[c] = h->xchg_ready[c];
        }
      }
    }
  }
  P.dense_S = direct ? R->big.A : nullptr;
  P.dense_ld = direct ? R->big.ld : 0;

  // small windows solved here alone: iteration 0 in fewer launches (below), S zeroed by the kernels (in the loop)
  const bool short_start = small && !sharded && !direct && P.n_long == 0 && P.n_act_poses > 0 && o.report_time_breakdown == 0;
  // iteration 0: cost at the starting point, fixed cost, Jacobi scaling
  HIPBA_HIP(launch_cost(P, lm, 0, n_act_blk + (short_start ? P.n_blk_cc : 0), 0, 1, false, s));
  if (!short_start) HIPBA_HIP(launch_cost(P, lm, n_act_blk, P.n_blk_cc, 0, 1, false, s));
  if (sharded) {
    HIPBA_HIP(launch_window_init(P, lm, kReduce, 0, s));
    // fixed cost belongs to the rank that owns the pose-only blocks; every rank evaluates the same CC
    // list only if it was given one, so the sum is what the caller sharded
    AR(P.wsum, 3, HIPBA_REDUCE_SUM);
    HIPBA_HIP(launch_window_init(P, lm, kDecide, 0, s));
  } else {
    HIPBA_HIP(launch_window_init(P, lm, kFused, short_start ? 1 : 0, s));
  }
  if (short_start) {
    // two launches: unscaled camera blocks beside the landmarks' column norms, then the pose scaling with the
    // reduced system's initialisation for the first pass (S and b were zeroed by k_window_init)
    HIPBA_HIP(launch_first_blocks(P, lm, s));
    HIPBA_HIP(launch_scale_init_reduced(P, lm, lm.jacobi_scaling, s));
  } else {
    HIPBA_HIP(launch_camera_blocks(P, lm, 0, 0, s));
    if (sharded && P.n_act_poses) { AR(P.U, 36 * (size_t)P.n_act_poses, HIPBA_REDUCE_SUM); AR(P.gc, 6 * (size_t)P.n_act_poses, HIPBA_REDUCE_SUM); }
    HIPBA_HIP(launch_make_scale_c(P, lm.jacobi_scaling, s));
    HIPBA_HIP(launch_point_colnorm(P, lm, s));
  }

  // optional per-phase device times (FullReport's evaluation / linear-solver split)
  const bool timing = o.report_time_breakdown != 0;
  if (timing && !h->phase_ev_ready) {
    for (auto& slot_ev : h->phase_ev)
      for (auto& ev : slot_ev)
        if (!ev) HIPBA_HIP(hipEventCreate(&ev));
    h->phase_ev_ready = true;
  }
  double t_res = 0.0, t_jac = 0.0, t_lin = 0.0;
  int collected = 0;  // iterations whose timestamps have been read
  auto collect = [&](int upto) -> int {  // the events of iterations <= upto have completed
    for (; collected < upto; ++collected) {
      hipEvent_t* ev = h->phase_ev[(collected + 1) & 3];
      float a = 0.f, b = 0.f, c = 0.f;
      HIPBA_HIP(hipEventElapsedTime(&a, ev[0], ev[1]));
      HIPBA_HIP(hipEventElapsedTime(&b, ev[1], ev[2]));
      HIPBA_HIP(hipEventElapsedTime(&c, ev[2], ev[3]));
      t_jac += 1e-3 * a; t_lin += 1e-3 * b; t_res += 1e-3 * c;
    }
    return HIPBA_OK;
  };
#define PHASE(k)                                                              \
  do {                                                                        \
    if (timing) HIPBA_HIP(hipEventRecord(h->phase_ev[iter & 3][(k)], s));     \
  } while (0)

  // small windows: the control kernel of a pass zeroes S for the next one (see k_window_control)
  const bool control_zeroes_S = small && !sharded && !direct;
  // development switch: HIPBA_SPLIT_LINEARIZE=1 keeps the camera blocks and the elimination in two launches (A/B timing)
  static const bool split_linearize = getenv("HIPBA_SPLIT_LINEARIZE") != nullptr;
  int iterations_run = 0;
  // (a window with a few thousand observations keeps the GPU busy for longer than the host needs to see a count
  // and queue a pass: one pass of lag is enough there and costs one idle pass at the end instead of two)
  const int lag = (small && !sharded && P.n_windows <= 8) ? (P.n_act_obs >= 8000 ? 1 : 2) : 0;
  // One small window: no event behind every pass (a packet of its own on the stream, and a wait in the runtime on
  // this side); the host reads the word the control kernel stores into the pinned ring, tagged with the pass number.
  const bool spin_poll = lag > 0 && P.n_windows == 1 && !timing;
  if (spin_poll) for (int i = 0; i < 4; ++i) h->h_count[i] = 0;  // (nothing of an earlier solve is in flight: each ends in a synchronize)
  // the control kernel of pass `want` stores (pass << 1) | active straight into the ring: waits for that word and
  // returns 1 (still active), 0 (stopped) or a negative error code
  auto pass_word = [&](int want) -> int {
    volatile int* word = h->h_count + (want & 3);
    double deadline = 0.0;
    for (long long spins = 0; (*word >> 1) != want; ++spins) {
      __builtin_ia32_pause();
      if ((spins & 0xfff) == 0xfff) {
        const double now = now_s();
        if (deadline == 0.0) deadline = now + 10.0;
        else if (now > deadline) {
          // ten seconds for a pass of a small window: something is wrong on the device; let the runtime say what
          HIPBA_HIP(hipStreamSynchronize(s));
          if ((*word >> 1) != want) return fail(HIPBA_ERR_INTERNAL, "pass %d of the solve never reported back", want);
        }
      }
    }
    return *word & 1;
  };
  // With one pass of lag the host does not queue a whole pass ahead but only its first launch (the linearisation,
  // the longest one but the solve): the word of the pass before is there by the time that launch has been queued, and
  // the rest of the pass follows while it runs.  A window that has stopped then costs one idle launch, not four.
  const bool split_pass = spin_poll && lag == 1;
  for (int iter = 1; iter <= std::max(lm.max_num_iterations, 0); ++iter) {
    PHASE(0);
    // S is zeroed first so that, unsharded, the camera-block launch can initialise the reduced system too
    if (direct) {
      if (!R->S_fully_written || sharded) HIPBA_HIP(big_zero_lower(P, R->big, env_ready, s));
    } else if (R->S_doubles && !(R->S_fully_written && !sharded) && !(control_zeroes_S && iter > 1) && !short_start) {
      HIPBA_HIP(hipMemsetAsync(P.S, 0, sizeof(double) * (size_t)R->S_doubles, s));
    }
    const bool fuse_init = iter > 1 && !sharded;
    // small windows from the second pass on: S and b were left zeroed by the control kernel, and one launch forms
    // the camera blocks, initialises the reduced system and eliminates the landmarks
    const bool one_launch = control_zeroes_S && iter > 1 && !timing && !split_linearize && P.n_act_poses > 0 && !P.prec;
    if (one_launch) {
      HIPBA_HIP(launch_small_linearize(P, lm, small_ncam, schur_parts, s));
    } else if (iter > 1) {
      HIPBA_HIP(launch_camera_blocks(P, lm, 1, fuse_init ? 1 : 0, s));
      if (sharded && P.n_act_poses) { AR(P.U, 36 * (size_t)P.n_act_poses, HIPBA_REDUCE_SUM); AR(P.gc, 6 * (size_t)P.n_act_poses, HIPBA_REDUCE_SUM); }
    }
    PHASE(1);
    if (!fuse_init && !(short_start && iter == 1)) HIPBA_HIP(launch_init_reduced(P, lm, owner, s));
    if (!one_launch) HIPBA_HIP(launch_point_schur(P, lm, small_ncam, schur_parts, s));
    if (sharded) {
      const bool piped = direct && feed.n_chunks > 0 && solver == HIPBA_LINEAR_DENSE_CHOLESKY;
      if (direct && !piped) {
        HIPBA_HIP(big_env_tiles(P, R->big, 0, n_env_tiles, P.S, kTilesPack, s));
        AR(P.S, (size_t)n_env_tiles * kCholNB * kCholNB, HIPBA_REDUCE_SUM);
        HIPBA_HIP(big_env_tiles(P, R->big, 0, n_env_tiles, P.S, kTilesUnpack, s));
      } else if (!direct && R->S_doubles) AR(P.S, (size_t)R->S_doubles, HIPBA_REDUCE_SUM);
      if (P.n_act_poses) AR(P.b, 6 * (size_t)P.n_act_poses, HIPBA_REDUCE_SUM);
      HIPBA_HIP(launch_window_gradient(P, lm, kReduce, s));
      AR(P.wsum, 1, HIPBA_REDUCE_MAX);
      AR(P.wsum + 1, 1, HIPBA_REDUCE_SUM);
      HIPBA_HIP(launch_window_gradient(P, lm, kDecide, s));
      if (piped) {
        // the matrix last: its chunks are summed on the exchange stream while this stream factors what has arrived
        HIPBA_HIP(big_env_tiles(P, R->big, 0, n_env_tiles, P.S, kTilesTake, s));
        HIPBA_HIP(hipEventRecord(h->xchg_packed, s));
        HIPBA_HIP(hipStreamWaitEvent(h->xchg_stream, h->xchg_packed, 0));
        for (int c = 0; c < feed.n_chunks; ++c) {
          const size_t t0 = (size_t)feed.tile_begin[c], t1 = (size_t)feed.tile_begin[c + 1];
          if (comm->allreduce(comm->ctx, P.S + t0 * kCholNB * kCholNB, (t1 - t0) * kCholNB * kCholNB, HIPBA_REDUCE_SUM,
                              (void*)h->xchg_stream)) {
            // the chunks already queued on the exchange stream still read the packed matrix: let them finish before
            // the caller gets its buffers back
            (void)hipStreamSynchronize(h->xchg_stream);
            return fail(HIPBA_ERR_COMM, "all-reduce callback failed");
          }
          HIPBA_HIP(hipEventRecord(feed.ready[c], h->xchg_stream));
        }
      }
    }
    if (split_pass && iter >= 2) {
      const int still = pass_word(iter - 1);
      if (still < 0) return still;
      if (still == 0) break;
    }
    if (small && !sharded) {
      HIPBA_HIP(launch_window_solve_small(P, lm, max_n, 1, iter == 1 ? 1 : 2, s));  // also forms the candidate poses
    } else {
      if (!sharded) HIPBA_HIP(launch_window_gradient(P, lm, kFused, s));
      if (small) {
        // sharded small window: gradient already decided; the LDS solver repeats the (idempotent) check
        HIPBA_HIP(launch_window_solve_small(P, lm, max_n, 0, 0, s));
      } else if (solver == HIPBA_LINEAR_PCG) {
        HIPBA_HIP(big_solve_pcg(P, R->big, o.pcg_max_iterations, o.pcg_tolerance, s));
      } else {
        HIPBA_HIP(big_solve_cholesky(P, R->big, env_ready ? R->env_first.data() : nullptr, s,
                                     sharded && direct && feed.n_chunks > 0 ? &feed : nullptr));
      }
    }
    // candidate poses first: the back-substitution then evaluates the candidate cost in the same pass over the
    // observations (kept apart when the caller wants the residual evaluation timed on its own)
    if (!(small && !sharded)) HIPBA_HIP(launch_pose_update(P, s));
    const bool cost_in_backsub = !timing && backsub_can_carry_cost(P);
    HIPBA_HIP(launch_backsub(P, lm, cost_in_backsub, s));
    PHASE(2);
    if (!cost_in_backsub) HIPBA_HIP(launch_cost(P, lm, 0, n_act_blk, 1, 0, true, s));
    PHASE(3);
    if (sharded) {
      HIPBA_HIP(launch_window_control(P, lm, kReduce, nullptr, 0, 0, s));
      AR(P.wsum, 5, HIPBA_REDUCE_SUM);
      HIPBA_HIP(launch_window_control(P, lm, kDecide, nullptr, 0, 0, s));
    } else {
      // a single window reports "still active" itself; batches are counted by k_count_active below
      HIPBA_HIP(launch_window_control(P, lm, kFused, P.n_windows == 1 ? h->h_count + (iter & 3) : nullptr, control_zeroes_S ? 1 : 0,
                                      spin_poll ? iter << 1 : 0, s));
    }
    iterations_run = iter;
    // "windows still active" travels to the host through a ring of pinned words.  Small windows: the host looks
    // at the count of iteration iter - 2 before queueing iter + 1, so the queue never runs dry between
    // iterations (every kernel returns at once for a window that has stopped; at most two idle iterations are
    // queued).  Large windows and big batches poll every iteration: an idle pass there is not free.
    const int slot = iter & 3;
    // (the kernels store the word straight into the pinned ring, which the device reaches at the same address: no
    // copy is queued behind them)
    if (sharded || P.n_windows != 1) HIPBA_HIP(launch_count_active(P, h->h_count + slot, s));
    if (!spin_poll) HIPBA_HIP(hipEventRecord(h->poll_ev[slot], s));
    if (iter - lag >= 1 && !split_pass) {
      const int seen = (iter - lag) & 3;
      if (spin_poll) {
        const int still = pass_word(iter - lag);
        if (still < 0) return still;
        if (still == 0) break;
      } else {
        HIPBA_HIP(hipEventSynchronize(h->poll_ev[seen]));
        if (timing) { const int rc = collect(iter - lag); if (rc) return rc; }
        if (h->h_count[seen] == 0) break;
      }
    }
  }
#undef AR
#undef PHASE
  // summaries
  std::vector<WinState> ws(P.n_windows);
  R->tail_valid = false;
  const size_t tail_bytes = R->tail_span;
  bool tail = fetch_tail && tail_bytes > 0 && tail_bytes <= ((size_t)1 << 20);
  if (tail && h->staging_bytes < tail_bytes) {
    // the staging buffer has to grow: nothing queued may still be reading the old one (the upload came from it)
    HIPBA_HIP(hipStreamSynchronize(s));
    tail = ensure_staging(h, tail_bytes) == HIPBA_OK;
  }
  if (tail) {
    // one copy, one round trip: everything the caller will read sits at the start of the arena (build_resident)
    char* st = static_cast<char*>(h->staging);
    const char* base = static_cast<const char*>(R->arena);
    HIPBA_HIP(hipMemcpyAsync(st, R->arena, tail_bytes, hipMemcpyDeviceToHost, s));
    R->tail_ws = reinterpret_cast<const WinState*>(st + (reinterpret_cast<const char*>(P.ws) - base));
    for (int k = 0; k < 2; ++k) {
      R->tail_poses[k] = reinterpret_cast<const double*>(st + (reinterpret_cast<const char*>(P.poses[k]) - base));
      R->tail_points[k] = reinterpret_cast<const double*>(st + (reinterpret_cast<const char*>(P.points[k]) - base));
    }
    HIPBA_HIP(hipStreamSynchronize(s));
    memcpy(ws.data(), R->tail_ws, sizeof(WinState) * P.n_windows);
    R->tail_valid = true;
    R->d2h_bytes = (long long)tail_bytes;
  } else {
    HIPBA_HIP(hipStreamSynchronize(s));
    HIPBA_HIP(hipMemcpy(ws.data(), P.ws, sizeof(WinState) * P.n_windows, hipMemcpyDeviceToHost));
    R->d2h_bytes = (long long)sizeof(WinState) * P.n_windows;
  }
  if (timing) { const int rc = collect(iterations_run); if (rc) return rc; }
  const double t1 = now_s();

  std::vector<hipba_iteration> log;
  bool want_log = o.minimizer_progress_to_stdout != 0;
  for (int w = 0; w < P.n_windows && !want_log; ++w) want_log = summaries[w].iteration_log != nullptr;
  if (want_log && P.log_cap > 0) {
    log.resize((size_t)P.log_cap * P.n_windows);
    HIPBA_HIP(hipMemcpy(log.data(), P.log, sizeof(hipba_iteration) * log.size(), hipMemcpyDeviceToHost));
    R->d2h_bytes += (long long)(sizeof(hipba_iteration) * log.size());
  }
  for (int w = 0; w < P.n_windows; ++w)
    if (ws[w].handoff_timeouts)
      return fail(HIPBA_ERR_INTERNAL, "reduced solve: a hand-off between workgroups timed out (window %d); the device "
                                      "was too busy for the one-launch pipeline -- the result is not reported", w);
  for (int w = 0; w < P.n_windows; ++w) {
    hipba_summary& sm = summaries[w];
    hipba_iteration* user_log = sm.iteration_log;
    const int user_cap = sm.iteration_log_capacity;
    const WinState& st = ws[w];
    int term = st.termination;
    if (st.active) term = HIPBA_NO_CONVERGENCE;  // loop ended on max_num_iterations on the host side
    sm.termination_type = term;
    sm.num_iterations = st.iteration + 1;
    // Ceres logs no iteration at all when there is nothing to optimise (no non-constant block in any residual: it
    // reports convergence at the fixed cost without running the minimizer) or when the very first evaluation fails
    // (a rank of a sharded solve cannot tell the first case from its own shard: it reports what the job did)
    if ((!sharded && (R->meta[w].n_act_poses + R->meta[w].n_act_points == 0 || R->meta[w].n_act_obs == 0)) ||
        (term == HIPBA_FAILURE && st.iteration == 0 && st.num_linear_solves == 0))
      sm.num_iterations = 0;
    sm.num_successful_steps = st.num_successful;
    sm.num_unsuccessful_steps = st.num_unsuccessful;
    sm.num_linear_solves = st.num_linear_solves;
    sm.initial_cost = st.initial_cost;
    sm.final_cost = st.x_cost + st.fixed_cost;
    sm.fixed_cost = st.fixed_cost;
    sm.num_active_poses = R->meta[w].n_act_poses;
    sm.num_active_points = R->meta[w].n_act_points;
    sm.num_active_observations = R->meta[w].n_act_obs;
    sm.time_solve_s = t1 - t0;
    sm.linear_solver_used = solver;
    sm.time_residual_evaluation_s = t_res;
    sm.time_jacobian_evaluation_s = t_jac;
    sm.time_linear_solver_s = t_lin;
    sm.bytes_host_to_device = 0;
    sm.bytes_device_to_host = R->d2h_bytes + 4ll * sm.num_linear_solves;  // + the 4-byte activity poll of every pass
    if (user_log && !log.empty()) {
      // (a capacity of zero or less copies nothing)
      const int n = std::min(std::min(sm.num_iterations, user_cap), P.log_cap);
      if (n > 0) memcpy(user_log, log.data() + (size_t)w * P.log_cap, sizeof(hipba_iteration) * (size_t)n);
    }
    if (o.minimizer_progress_to_stdout && !log.empty()) {
      if (P.n_windows > 1) printf("window %d\n", w);
      print_progress(log.data() + (size_t)w * P.log_cap, std::min(sm.num_iterations, P.log_cap));
      static const char* names[] = {"CONVERGENCE", "NO_CONVERGENCE", "FAILURE"};
      printf("Termination: %s  initial cost %.6e  final cost %.6e  iterations %d  (%.3f ms)\n",
             names[term < 0 || term > 2 ? 2 : term], sm.initial_cost, sm.final_cost, sm.num_iterations,
             1e3 * sm.time_solve_s);
    }
  }
  return HIPBA_OK;
}

// current parameters of every window, into caller arrays (window w: probs[w].poses / points)
int download_all(hipba_handle_s* h, hipba_device_problem_s* R, hipba_problem* probs, const hipba_summary* summaries) {
  const DevP& P = R->P;
  std::vector<WinState> ws(P.n_windows);
  std::vector<double> poses[2], points[2];
  const double* hp[2] = {nullptr, nullptr};
  const double* hq[2] = {nullptr, nullptr};
  if (R->tail_valid) {
    // the solve has already brought everything into the pinned staging buffer
    memcpy(ws.data(), R->tail_ws, sizeof(WinState) * P.n_windows);
    for (int k = 0; k < 2; ++k) { hp[k] = R->tail_poses[k]; hq[k] = R->tail_points[k]; }
  } else {
  HIPBA_HIP(hipMemcpy(ws.data(), P.ws, sizeof(WinState) * P.n_windows, hipMemcpyDeviceToHost));
  R->d2h_bytes += (long long)sizeof(WinState) * P.n_windows;
  for (int k = 0; k < 2; ++k) {
    if (P.n_windows == 1 && k != ws[0].sel) continue;  // one window: only the buffer that holds x
    poses[k].resize(std::max<size_t>(7 * (size_t)P.n_poses, 7));
    points[k].resize(std::max<size_t>(3 * (size_t)P.n_pm_points, 3));
    if (P.n_poses) HIPBA_HIP(hipMemcpyAsync(poses[k].data(), P.poses[k], sizeof(double) * 7 * P.n_poses, hipMemcpyDeviceToHost, h->stream));
    if (P.n_pm_points) HIPBA_HIP(hipMemcpyAsync(points[k].data(), P.points[k], sizeof(double) * 3 * P.n_pm_points, hipMemcpyDeviceToHost, h->stream));
    R->d2h_bytes += 8ll * (7ll * P.n_poses + 3ll * P.n_pm_points);
    hp[k] = poses[k].data(); hq[k] = points[k].data();
  }
  HIPBA_HIP(hipStreamSynchronize(h->stream));
  }
  for (int w = 0; w < P.n_windows; ++w) {
    if (summaries && summaries[w].termination_type == HIPBA_FAILURE) continue;  // Ceres: solution not usable
    const int sel = ws[w].sel;
    const WinMeta& m = R->meta[w];
    if (m.n_poses) memcpy(probs[w].poses, hp[sel] + 7ll * m.pose_base, sizeof(double) * 7 * m.n_poses);
    for (int i = 0; i < m.n_points; ++i) {
      const int dp = m.dp_of[i];
      if (dp < P.n_pm_points) {
        probs[w].points[3 * i] = hq[sel][3ll * dp];
        probs[w].points[3 * i + 1] = hq[sel][3ll * dp + 1];
        probs[w].points[3 * i + 2] = hq[sel][3ll * dp + 2];
      }
    }
  }
  return HIPBA_OK;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int hipba_version(void) { return HIPBA_VERSION; }

const char* hipba_last_error(void) { return g_last_error.c_str(); }

void hipba_options_default(hipba_options* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->loss = HIPBA_LOSS_NONE;
  o->loss_scale = 1.0;
  o->linear_solver = HIPBA_LINEAR_AUTO;
  o->jacobi_scaling = 1;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->pcg_max_iterations = 500;
  o->pcg_tolerance = 1e-12;
  o->minimizer_progress_to_stdout = 0;
}

int hipba_device_count(int* count) {
  return guarded([&]() -> int {
    if (!count) return fail(HIPBA_ERR_INVALID_ARGUMENT, "count is null");
    *count = 0;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
      return fail(HIPBA_ERR_NO_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    *count = n;
    return HIPBA_OK;
  });
}

int hipba_create(int device, hipba_handle* out) {
  return guarded([&]() -> int {
    if (!out) return fail(HIPBA_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
      return fail(HIPBA_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU path",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(HIPBA_ERR_INVALID_ARGUMENT, "device %d out of range (0..%d)", device, count - 1);
    HIPBA_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPBA_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
      return fail(HIPBA_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    // kernel attributes are per device: set them for this handle's device whatever other handles have done
    HIPBA_HIP(configure_lm_kernels());
    HIPBA_HIP(configure_dense_kernels());
    double* eval_scratch = nullptr;
    HIPBA_HIP(scratch_on_device(device, &eval_scratch));
    auto* h = new hipba_handle_s();
    h->device = device;
    h->eval_scratch = eval_scratch;
    e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_count, 64, hipHostMallocDefault);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->poll_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) { delete h; return fail(HIPBA_ERR_HIP, "handle creation: %s", hipGetErrorString(e)); }
    *out = h;
    return HIPBA_OK;
  });
}

int hipba_destroy(hipba_handle h) {
  return guarded([&]() -> int {
    if (!h) return HIPBA_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->arena) (void)hipFree(h->arena);
    if (h->staging) (void)hipHostFree(h->staging);
    if (h->h_count) (void)hipHostFree(h->h_count);
    for (int i = 0; i < 4; ++i) if (h->poll_ev[i]) (void)hipEventDestroy(h->poll_ev[i]);
    for (auto& slot_ev : h->phase_ev)
      for (auto& ev : slot_ev) if (ev) (void)hipEventDestroy(ev);
    if (h->xchg_stream) (void)hipStreamSynchronize(h->xchg_stream);
    if (h->xchg_packed) (void)hipEventDestroy(h->xchg_packed);
    for (auto& ev : h->xchg_ready) if (ev) (void)hipEventDestroy(ev);
    if (h->xchg_stream) (void)hipStreamDestroy(h->xchg_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return HIPBA_OK;
  });
}

void* hipba_stream(hipba_handle h) { return h ? (void*)h->stream : nullptr; }

// A window that the batched kernels cannot take (they keep every window's reduced system in LDS and every landmark
// inside one 256-observation block): more variable poses than the in-LDS solver holds, or a landmark with more
// observations than a block.  Judged from the caller's arrays alone, on the safe side: poses that turn out to be
// unobserved still count.  Malformed input is left to validate_problem.
static bool window_needs_the_large_solver(const hipba_problem& p) {
  if (p.n_poses <= 0 || p.n_obs <= 0 || !p.obs_point || p.n_points <= 0) return false;
  int variable = 0;
  for (int i = 0; i < p.n_poses; ++i) variable += (p.pose_const && p.pose_const[i]) ? 0 : 1;
  if (6 * variable > kSmallMaxN) return true;
  if (p.n_obs <= kBlock) return false;
  std::vector<int> degree((size_t)p.n_points, 0);
  for (long long k = 0; k < p.n_obs; ++k) {
    const int q = p.obs_point[k];
    if (q < 0 || q >= p.n_points) return false;
    if (++degree[q] > kBlock) return true;
  }
  return false;
}

int hipba_solve_batched(hipba_handle h, int32_t n_windows, hipba_problem* problems, const hipba_options* options,
                        hipba_summary* summaries) {
  return guarded([&]() -> int {
    if (!h || !problems || !summaries || n_windows <= 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    if (n_windows > 1) {
      // The reference sets no limit on a window (optimizer.cpp:25-51).  Windows too large for the batched kernels go
      // through the single-window path one after another, the rest stay one batch; the results are what separate
      // calls would give.
      std::vector<int> large, regular;
      for (int w = 0; w < n_windows; ++w) (window_needs_the_large_solver(problems[w]) ? large : regular).push_back(w);
      if (!large.empty()) {
        if (!regular.empty()) {
          std::vector<hipba_problem> sub(regular.size());
          std::vector<hipba_summary> sub_summary(regular.size());
          for (size_t i = 0; i < regular.size(); ++i) { sub[i] = problems[regular[i]]; sub_summary[i] = summaries[regular[i]]; }
          const int rc = hipba_solve_batched(h, (int32_t)regular.size(), sub.data(), options, sub_summary.data());
          if (rc) return rc;
          for (size_t i = 0; i < regular.size(); ++i) summaries[regular[i]] = sub_summary[i];
        }
        for (int w : large) {
          const int rc = hipba_solve_batched(h, 1, problems + w, options, summaries + w);
          if (rc) return rc;
        }
        return HIPBA_OK;
      }
    }
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    HIPBA_HIP(hipSetDevice(h->device));
    const double t0 = now_s();
    hipba_device_problem_s* raw = nullptr;
    int rc = build_resident(h, n_windows, problems, true, std::max(o.max_num_iterations, 0) + 1, false, &raw);
    if (rc) return rc;
    std::unique_ptr<hipba_device_problem_s, ResidentDeleter> owner(raw);
    hipba_device_problem_s* R = raw;
    const double t1 = now_s();
    rc = run_lm(h, R, o, nullptr, summaries, true);
    if (rc) (void)hipStreamSynchronize(h->stream);  // nothing queued may outlive the buffers of this call
    if (!rc) rc = download_all(h, R, problems, summaries);
    const long long upload_bytes = R->upload_bytes, d2h_bytes = R->d2h_bytes + 4ll * (n_windows > 0 ? summaries[0].num_linear_solves : 0);
    owner.reset();
    const double t2 = now_s();
    if (!rc) {
      for (int w = 0; w < n_windows; ++w) {
        summaries[w].time_setup_s = t1 - t0;
        summaries[w].time_total_s = t2 - t0;
        summaries[w].bytes_host_to_device = upload_bytes;
        summaries[w].bytes_device_to_host = d2h_bytes;
      }
    }
    return rc;
  });
}

int hipba_solve(hipba_handle h, hipba_problem* problem, const hipba_options* options, hipba_summary* summary) {
  return hipba_solve_batched(h, 1, problem, options, summary);
}

int hipba_upload(hipba_handle h, int32_t n_windows, const hipba_problem* problems, hipba_device_problem* out) {
  return guarded([&]() -> int {
    if (!h || !problems || !out || n_windows <= 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    HIPBA_HIP(hipSetDevice(h->device));
    hipba_device_problem_s* R = nullptr;
    // resident problems keep room for Ceres' production iteration budget (optimizer.cpp:58)
    int rc = build_resident(h, n_windows, problems, false, 301, false, &R);
    if (rc) return rc;
    *out = R;
    return HIPBA_OK;
  });
}

int hipba_pose_order(const hipba_problem* p, int32_t* order) {
  return guarded([&]() -> int {
    if (!p || !order) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    int rc = validate_problem(*p, 0);
    if (rc) return rc;
    const int np = p->n_poses;
    // variable poses that see a variable landmark, numbered in pose order
    std::vector<int> node(np, -1);
    std::vector<long long> start((size_t)p->n_points + 1, 0);
    for (long long k = 0; k < p->n_obs; ++k) start[p->obs_point[k] + 1]++;
    for (int i = 0; i < p->n_points; ++i) start[i + 1] += start[i];
    std::vector<int> by_point((size_t)std::max<long long>(p->n_obs, 1));
    {
      std::vector<long long> cur(start.begin(), start.end() - 1);
      for (long long k = 0; k < p->n_obs; ++k) by_point[cur[p->obs_point[k]]++] = p->obs_pose[k];
    }
    int n_nodes = 0;
    for (int c = 0; c < np; ++c)
      if (!(p->pose_const && p->pose_const[c])) node[c] = n_nodes++;
    PoseGraph graph(std::max(n_nodes, 1));
    std::vector<int> seg;
    for (int q = 0; q < p->n_points; ++q) {
      if (p->point_const && p->point_const[q]) continue;
      seg.clear();
      for (long long o = start[q]; o < start[q + 1]; ++o) {
        const int g = node[by_point[o]];
        if (g >= 0 && std::find(seg.begin(), seg.end(), g) == seg.end()) seg.push_back(g);
      }
      for (size_t i = 0; i < seg.size(); ++i)
        for (size_t j = i + 1; j < seg.size(); ++j) graph.connect(seg[i], seg[j]);
    }
    std::vector<int> ord(n_nodes);
    std::iota(ord.begin(), ord.end(), 0);
    if (n_nodes > 1) {
      const int n_tiles = (6 * n_nodes + kCholNB - 1) / kCholNB;
      double cost_id = 0.0, cost_rcm = 0.0;
      tile_envelope(graph, ord, kCholNB, n_tiles, &cost_id);
      std::vector<int> rcm = reverse_cuthill_mckee(graph);
      tile_envelope(graph, rcm, kCholNB, n_tiles, &cost_rcm);
      if (cost_rcm < 0.9 * cost_id) ord.swap(rcm);
    }
    std::vector<int> pose_of(n_nodes);
    for (int c = 0; c < np; ++c)
      if (node[c] >= 0) pose_of[node[c]] = c;
    int k = 0;
    for (int i = 0; i < n_nodes; ++i) order[k++] = pose_of[ord[i]];
    for (int c = 0; c < np; ++c)
      if (node[c] < 0) order[k++] = c;
    return HIPBA_OK;
  });
}

int hipba_upload_shard(hipba_handle h, const hipba_problem* shard, const int32_t* pose_order, hipba_device_problem* out) {
  return guarded([&]() -> int {
    if (!h || !shard || !out) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    HIPBA_HIP(hipSetDevice(h->device));
    hipba_device_problem_s* R = nullptr;
    int rc = build_resident(h, 1, shard, false, 301, true, &R, pose_order);
    if (rc) return rc;
    *out = R;
    return HIPBA_OK;
  });
}

int hipba_reset(hipba_handle h, hipba_device_problem dp) {
  return guarded([&]() -> int {
    if (!h || !dp) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    HIPBA_HIP(hipSetDevice(h->device));
    const DevP& P = dp->P;
    const size_t pb = sizeof(double) * std::max<size_t>(7 * (size_t)P.n_poses, 7);
    const size_t qb = sizeof(double) * std::max<size_t>(3 * (size_t)P.n_points, 3);
    for (int k = 0; k < 2; ++k) {
      HIPBA_HIP(hipMemcpyAsync(P.poses[k], dp->poses0, pb, hipMemcpyDeviceToDevice, h->stream));
      HIPBA_HIP(hipMemcpyAsync(P.points[k], dp->points0, qb, hipMemcpyDeviceToDevice, h->stream));
    }
    HIPBA_HIP(hipMemsetAsync(P.ws, 0, sizeof(WinState) * P.n_windows, h->stream));
    HIPBA_HIP(hipStreamSynchronize(h->stream));
    return HIPBA_OK;
  });
}

int hipba_solve_device(hipba_handle h, hipba_device_problem dp, const hipba_options* options, hipba_summary* summaries) {
  return guarded([&]() -> int {
    if (!h || !dp || !summaries) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    if (o.max_num_iterations + 1 > dp->P.log_cap) {
      // the log of a resident problem has a fixed capacity; longer runs simply stop logging
    }
    HIPBA_HIP(hipSetDevice(h->device));
    const double t0 = now_s();
    int rc = run_lm(h, dp, o, nullptr, summaries);
    if (!rc)
      for (int w = 0; w < dp->P.n_windows; ++w) { summaries[w].time_setup_s = 0.0; summaries[w].time_total_s = now_s() - t0; }
    return rc;
  });
}

int hipba_solve_device_sharded(hipba_handle h, hipba_device_problem dp, const hipba_comm* comm,
                               const hipba_options* options, hipba_summary* summary) {
  return guarded([&]() -> int {
    if (!h || !dp || !summary || !comm) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    HIPBA_HIP(hipSetDevice(h->device));
    const double t0 = now_s();
    int rc = run_lm(h, dp, o, comm, summary);
    if (!rc) { summary->time_setup_s = 0.0; summary->time_total_s = now_s() - t0; }
    return rc;
  });
}

int hipba_download(hipba_handle h, hipba_device_problem dp, int32_t window, double* poses, double* points) {
  return guarded([&]() -> int {
    if (!h || !dp) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    if (window < 0 || window >= dp->P.n_windows) return fail(HIPBA_ERR_INVALID_ARGUMENT, "window %d out of range", window);
    HIPBA_HIP(hipSetDevice(h->device));
    const DevP& P = dp->P;
    const WinMeta& m = dp->meta[window];
    WinState st;
    HIPBA_HIP(hipMemcpy(&st, P.ws + window, sizeof(WinState), hipMemcpyDeviceToHost));
    if (poses && m.n_poses)
      HIPBA_HIP(hipMemcpy(poses, P.poses[st.sel] + 7ll * m.pose_base, sizeof(double) * 7 * m.n_poses, hipMemcpyDeviceToHost));
    if (points && m.n_points) {
      std::vector<double> all(3 * (size_t)P.n_points);
      HIPBA_HIP(hipMemcpy(all.data(), P.points[st.sel], sizeof(double) * all.size(), hipMemcpyDeviceToHost));
      for (int i = 0; i < m.n_points; ++i) {
        const int d = m.dp_of[i];
        points[3 * i] = all[3ll * d]; points[3 * i + 1] = all[3ll * d + 1]; points[3 * i + 2] = all[3ll * d + 2];
      }
    }
    return HIPBA_OK;
  });
}

int hipba_free_problem(hipba_handle h, hipba_device_problem dp) {
  return guarded([&]() -> int {
    if (!dp) return HIPBA_OK;
    if (h) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
    free_resident(dp);
    return HIPBA_OK;
  });
}

int hipba_evaluate(hipba_handle h, const hipba_problem* p, const hipba_options* options, int apply_loss, double* cost,
                   double* residuals, double* jac_pose, double* jac_point) {
  return guarded([&]() -> int {
    if (!h || !p) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    int rc = validate_problem(*p, 0);
    if (rc) return rc;
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    HIPBA_HIP(hipSetDevice(h->device));
    const long long n = p->n_obs;
    if (n == 0) { if (cost) *cost = 0.0; return HIPBA_OK; }
    const long long nb = (n + kBlock - 1) / kBlock;
    ArenaPlan plan;
    double *d_poses, *d_points, *d_uv, *d_res, *d_jc, *d_jp, *d_part, *d_cost;
    int *d_op, *d_oq;
    plan.add(&d_poses, 7 * (size_t)p->n_poses);
    plan.add(&d_points, 3 * (size_t)p->n_points);
    plan.add(&d_uv, 2 * (size_t)n);
    plan.add(&d_op, (size_t)n);
    plan.add(&d_oq, (size_t)n);
    plan.add(&d_res, 2 * (size_t)n);
    plan.add(&d_jc, 12 * (size_t)n);
    plan.add(&d_jp, 6 * (size_t)n);
    plan.add(&d_part, (size_t)nb);
    plan.add(&d_cost, 2);
    void* base = nullptr;
    HIPBA_HIP(hipMalloc(&base, plan.total));
    plan.bind(base);
    hipStream_t s = h->stream;
    auto body = [&]() -> int {
      HIPBA_HIP(hipMemcpyAsync(d_poses, p->poses, sizeof(double) * 7 * p->n_poses, hipMemcpyHostToDevice, s));
      HIPBA_HIP(hipMemcpyAsync(d_points, p->points, sizeof(double) * 3 * p->n_points, hipMemcpyHostToDevice, s));
      HIPBA_HIP(hipMemcpyAsync(d_uv, p->obs_uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
      HIPBA_HIP(hipMemcpyAsync(d_op, p->obs_pose, sizeof(int) * n, hipMemcpyHostToDevice, s));
      HIPBA_HIP(hipMemcpyAsync(d_oq, p->obs_point, sizeof(int) * n, hipMemcpyHostToDevice, s));
      const CameraK K{p->fx, p->fy, p->cx, p->cy, p->camera_model};
      HIPBA_HIP(launch_residual_jacobian(K, make_lm(o), apply_loss, n, p->n_poses, p->n_points, d_poses, d_points, d_op, d_oq, d_uv,
                                         residuals ? d_res : nullptr, jac_pose ? d_jc : nullptr,
                                         jac_point ? d_jp : nullptr, d_part, d_cost, s));
      if (residuals) HIPBA_HIP(hipMemcpyAsync(residuals, d_res, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s));
      if (jac_pose) HIPBA_HIP(hipMemcpyAsync(jac_pose, d_jc, sizeof(double) * 12 * n, hipMemcpyDeviceToHost, s));
      if (jac_point) HIPBA_HIP(hipMemcpyAsync(jac_point, d_jp, sizeof(double) * 6 * n, hipMemcpyDeviceToHost, s));
      if (cost) HIPBA_HIP(hipMemcpyAsync(cost, d_cost, sizeof(double), hipMemcpyDeviceToHost, s));
      HIPBA_HIP(hipStreamSynchronize(s));
      return HIPBA_OK;
    };
    rc = body();
    (void)hipFree(base);
    return rc;
  });
}

int hipba_evaluate_device(hipba_handle h, int32_t n_poses, const double* d_poses, int32_t n_points, const double* d_points,
                          int64_t n_obs, const int32_t* d_obs_pose, const int32_t* d_obs_point, const double* d_obs_uv,
                          double fx, double fy, double cx, double cy, int32_t camera_model, const hipba_options* options,
                          int apply_loss, double* d_cost, double* d_residuals, double* d_jac_pose, double* d_jac_point) {
  return guarded([&]() -> int {
    if (!h) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null handle");
    if (n_poses < 0 || n_points < 0 || n_obs < 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "negative size");
    if (n_obs > (int64_t)HIPBA_EVALUATE_DEVICE_MAX_OBS)
      return fail(HIPBA_ERR_UNSUPPORTED, "hipba_evaluate_device takes at most %lld observations per call (got %lld): "
                  "its scratch is a fixed array in the library; split the call or use hipba_evaluate",
                  (long long)HIPBA_EVALUATE_DEVICE_MAX_OBS, (long long)n_obs);
    if (camera_model != HIPBA_CAMERA_Z_FORWARD && camera_model != HIPBA_CAMERA_X_FORWARD)
      return fail(HIPBA_ERR_INVALID_ARGUMENT, "unknown camera model %d", camera_model);
    if (n_obs && (!d_poses || !d_points || !d_obs_pose || !d_obs_point || !d_obs_uv))
      return fail(HIPBA_ERR_INVALID_ARGUMENT, "null input array");
    if (!h->eval_scratch) return fail(HIPBA_ERR_HIP, "the handle has no scratch address");
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    HIPBA_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    if (n_obs == 0) {
      if (d_cost) HIPBA_HIP(hipMemsetAsync(d_cost, 0, sizeof(double), s));
      return HIPBA_OK;
    }
    const CameraK K{fx, fy, cx, cy, camera_model};
    // the partial sums go to this device's copy of the library's scratch array: nothing is allocated here
    HIPBA_HIP(launch_residual_jacobian(K, make_lm(o), apply_loss, n_obs, n_poses, n_points, d_poses, d_points, d_obs_pose,
                                       d_obs_point, d_obs_uv, d_residuals, d_jac_pose, d_jac_point,
                                       d_cost ? h->eval_scratch : nullptr, d_cost, s));
    return HIPBA_OK;
  });
}

int hipba_triangulate_stereo_device(hipba_handle h, const double* d_P_left, const double* d_P_right, int64_t n,
                                    const float* d_left_uv, const float* d_right_uv, const double* d_pose,
                                    double* d_points) {
  return guarded([&]() -> int {
    if (!h) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return HIPBA_OK;
    if (!d_P_left || !d_P_right || !d_left_uv || !d_right_uv || !d_points)
      return fail(HIPBA_ERR_INVALID_ARGUMENT, "null input array");
    HIPBA_HIP(hipSetDevice(h->device));
    HIPBA_HIP(launch_triangulate_stereo(d_P_left, d_P_right, n, d_left_uv, d_right_uv, d_pose, d_points, h->stream));
    return HIPBA_OK;
  });
}

int hipba_triangulate_stereo(hipba_handle h, const double P_left[12], const double P_right[12], int64_t n,
                             const float* left_uv, const float* right_uv, const double* pose, double* points) {
  return guarded([&]() -> int {
    if (!h) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "negative size");
    if (n == 0) return HIPBA_OK;
    if (!P_left || !P_right || !left_uv || !right_uv || !points) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null input array");
    HIPBA_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // one staging block, one copy each way: [P_left 12][P_right 12][pose 7 + 1][left 2n floats][right 2n floats] -> [3n]
    const size_t head = sizeof(double) * 32, uv_bytes = sizeof(float) * 2 * (size_t)n, out_bytes = sizeof(double) * 3 * (size_t)n;
    const size_t in_bytes = align_up(head + 2 * uv_bytes, 256);
    int rc = ensure_staging(h, std::max(in_bytes, out_bytes));
    if (rc) return rc;
    rc = ensure_arena(h, in_bytes + out_bytes);
    if (rc) return rc;
    char* st = static_cast<char*>(h->staging);
    double* hd = reinterpret_cast<double*>(st);
    for (int i = 0; i < 12; ++i) { hd[i] = P_left[i]; hd[12 + i] = P_right[i]; }
    for (int i = 0; i < 7; ++i) hd[24 + i] = pose ? pose[i] : 0.0;
    hd[31] = 0.0;
    memcpy(st + head, left_uv, uv_bytes);
    memcpy(st + head + uv_bytes, right_uv, uv_bytes);
    char* dv = static_cast<char*>(h->arena);
    HIPBA_HIP(hipMemcpyAsync(dv, st, head + 2 * uv_bytes, hipMemcpyHostToDevice, s));
    const double* d_head = reinterpret_cast<const double*>(dv);
    double* d_out = reinterpret_cast<double*>(dv + in_bytes);
    HIPBA_HIP(launch_triangulate_stereo(d_head, d_head + 12, n, reinterpret_cast<const float*>(dv + head),
                                        reinterpret_cast<const float*>(dv + head + uv_bytes), pose ? d_head + 24 : nullptr,
                                        d_out, s));
    HIPBA_HIP(hipMemcpyAsync(st, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    HIPBA_HIP(hipStreamSynchronize(s));
    memcpy(points, st, out_bytes);
    return HIPBA_OK;
  });
}

int hipba_se3_plus(hipba_handle h, int64_t n, const double* poses, const double* deltas, double* out) {
  return guarded([&]() -> int {
    if (!h) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || (n > 0 && (!poses || !deltas || !out))) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null array");
    if (n == 0) return HIPBA_OK;
    HIPBA_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const size_t pose_bytes = align_up(sizeof(double) * 7 * (size_t)n, 256), delta_bytes = align_up(sizeof(double) * 6 * (size_t)n, 256);
    int rc = ensure_staging(h, pose_bytes + delta_bytes);
    if (rc) return rc;
    rc = ensure_arena(h, 2 * pose_bytes + delta_bytes);
    if (rc) return rc;
    char* st = static_cast<char*>(h->staging);
    memcpy(st, poses, sizeof(double) * 7 * (size_t)n);
    memcpy(st + pose_bytes, deltas, sizeof(double) * 6 * (size_t)n);
    char* dv = static_cast<char*>(h->arena);
    HIPBA_HIP(hipMemcpyAsync(dv, st, pose_bytes + delta_bytes, hipMemcpyHostToDevice, s));
    double* d_out = reinterpret_cast<double*>(dv + pose_bytes + delta_bytes);
    HIPBA_HIP(launch_se3_plus(n, reinterpret_cast<const double*>(dv), reinterpret_cast<const double*>(dv + pose_bytes), d_out, s));
    HIPBA_HIP(hipMemcpyAsync(st, d_out, sizeof(double) * 7 * (size_t)n, hipMemcpyDeviceToHost, s));
    HIPBA_HIP(hipStreamSynchronize(s));
    memcpy(out, st, sizeof(double) * 7 * (size_t)n);
    return HIPBA_OK;
  });
}

int hipba_refine_pose(hipba_handle h, double fx, double fy, double cx, double cy, int32_t camera_model, int64_t n,
                      const double* points, const double* uv, const uint8_t* use, double pose[7],
                      const hipba_options* options, hipba_summary* summary) {
  return guarded([&]() -> int {
    if (!h || !pose) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 0 || (n > 0 && (!points || !uv))) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null input array");
    // one variable pose, the matches in use as constant landmarks with one observation each
    std::vector<double> pts, px;
    pts.reserve(3 * (size_t)n); px.reserve(2 * (size_t)n);
    for (int64_t i = 0; i < n; ++i) {
      if (use && !use[i]) continue;
      pts.insert(pts.end(), points + 3 * i, points + 3 * i + 3);
      px.insert(px.end(), uv + 2 * i, uv + 2 * i + 2);
    }
    const size_t m = px.size() / 2;
    if (m < 3) return fail(HIPBA_ERR_INVALID_ARGUMENT, "pose refinement needs at least 3 matches in use (got %zu)", m);
    if (m > (size_t)INT32_MAX) return fail(HIPBA_ERR_INVALID_ARGUMENT, "too many matches");
    std::vector<uint8_t> point_const(m, 1);
    std::vector<int32_t> obs_pose(m, 0), obs_point(m);
    for (size_t i = 0; i < m; ++i) obs_point[i] = (int32_t)i;
    hipba_problem p{};
    p.n_poses = 1; p.poses = pose; p.pose_const = nullptr;
    p.n_points = (int32_t)m; p.points = pts.data(); p.point_const = point_const.data();
    p.n_obs = (int64_t)m; p.obs_pose = obs_pose.data(); p.obs_point = obs_point.data(); p.obs_uv = px.data();
    p.fx = fx; p.fy = fy; p.cx = cx; p.cy = cy; p.camera_model = camera_model;
    hipba_summary local{};  // (iteration_log = null: the solve writes no log)
    return hipba_solve(h, &p, options, summary ? summary : &local);
  });
}

int hipba_pnp_ransac(hipba_handle h, double fx, double fy, double cx, double cy, int32_t camera_model, int64_t n,
                     const double* points, const double* uv, double pose[7], int32_t iterations,
                     double reprojection_error, double confidence, uint64_t seed, int32_t refine, uint8_t* inliers,
                     int64_t* n_inliers, int32_t* iterations_run) {
  return guarded([&]() -> int {
    if (!h || !pose || !points || !uv) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    if (!(confidence >= 0.0 && confidence <= 1.0)) return fail(HIPBA_ERR_INVALID_ARGUMENT, "confidence must lie in [0, 1]");
    if (n < 4) return fail(HIPBA_ERR_INVALID_ARGUMENT, "PnP by sample consensus needs at least 4 matches (got %lld)", (long long)n);
    if (iterations <= 0 || iterations > (1 << 20)) return fail(HIPBA_ERR_INVALID_ARGUMENT, "iterations out of range (%d)", iterations);
    if (!(reprojection_error > 0.0)) return fail(HIPBA_ERR_INVALID_ARGUMENT, "reprojection_error must be positive");
    if (camera_model != HIPBA_CAMERA_Z_FORWARD && camera_model != HIPBA_CAMERA_X_FORWARD)
      return fail(HIPBA_ERR_INVALID_ARGUMENT, "unknown camera model %d", camera_model);
    HIPBA_HIP(hipSetDevice(h->device));
    hipStream_t s = h->stream;
    // one staging block up: [prior 8][points 3n][uv 2n]; device work area behind it: counts, poses, result, mask
    const size_t up_bytes = sizeof(double) * (8 + 5 * (size_t)n);
    const size_t count_bytes = align_up(sizeof(int) * (size_t)iterations, 256);
    const size_t poses_bytes = align_up(sizeof(double) * 7 * (size_t)iterations, 256);
    const size_t mask_bytes = align_up((size_t)n, 256);
    const size_t up_al = align_up(up_bytes, 256);
    int rc = ensure_staging(h, std::max(up_bytes, 256 + mask_bytes));
    if (rc) return rc;
    rc = ensure_arena(h, up_al + count_bytes + poses_bytes + 256 + mask_bytes);
    if (rc) return rc;
    double* st = static_cast<double*>(h->staging);
    for (int i = 0; i < 7; ++i) st[i] = pose[i];
    st[7] = 0.0;
    memcpy(st + 8, points, sizeof(double) * 3 * (size_t)n);
    memcpy(st + 8 + 3 * (size_t)n, uv, sizeof(double) * 2 * (size_t)n);
    char* dv = static_cast<char*>(h->arena);
    HIPBA_HIP(hipMemcpyAsync(dv, st, up_bytes, hipMemcpyHostToDevice, s));
    const double* d_prior = reinterpret_cast<const double*>(dv);
    const double* d_pts = d_prior + 8;
    const double* d_uv = d_pts + 3 * (size_t)n;
    int* d_count = reinterpret_cast<int*>(dv + up_al);
    double* d_poses = reinterpret_cast<double*>(dv + up_al + count_bytes);
    double* d_out = reinterpret_cast<double*>(dv + up_al + count_bytes + poses_bytes);
    unsigned char* d_mask = reinterpret_cast<unsigned char*>(dv + up_al + count_bytes + poses_bytes + 256);
    const CameraK K{fx, fy, cx, cy, camera_model};
    // a quaternion of zeros is no rotation: the caller has no prior, the minimal samples are solved in closed form
    const int closed_form = pose[0] == 0.0 && pose[1] == 0.0 && pose[2] == 0.0 && pose[3] == 0.0;
    HIPBA_HIP(launch_pnp_ransac(K, n, d_pts, d_uv, d_prior, closed_form, iterations, seed, reprojection_error, confidence,
                                d_count, d_poses, d_out, d_mask, s));
    // result and mask come back in one copy (they are adjacent in the work area)
    HIPBA_HIP(hipMemcpyAsync(st, d_out, 256 + (size_t)n, hipMemcpyDeviceToHost, s));
    HIPBA_HIP(hipStreamSynchronize(s));
    const long long best = (long long)st[7];
    if (iterations_run) *iterations_run = (int32_t)st[8];
    bool finite = true;
    for (int i = 0; i < 7; ++i) finite = finite && std::isfinite(st[i]);
    if (best < 4 || !finite) {
      // no hypothesis with a consensus of at least the minimal sample (cv::solvePnPRansac returns false and leaves
      // rvec / tvec alone, estimator.cpp:28-29): the caller's pose stays, no inliers are reported
      if (inliers) memset(inliers, 0, (size_t)n);
      if (n_inliers) *n_inliers = 0;
      return HIPBA_OK;
    }
    for (int i = 0; i < 7; ++i) pose[i] = st[i];
    const unsigned char* mask = reinterpret_cast<const unsigned char*>(st) + 256;
    std::vector<uint8_t> keep(mask, mask + n);  // (the refinement below reuses the staging buffer)
    if (inliers) memcpy(inliers, keep.data(), (size_t)n);
    if (n_inliers) *n_inliers = best;
    if (refine) return hipba_refine_pose(h, fx, fy, cx, cy, camera_model, n, points, uv, keep.data(), pose, nullptr, nullptr);
    return HIPBA_OK;
  });
}

int hipba_reduced_solve_flops(hipba_device_problem dp, double* flops_dense, double* flops_envelope) {
  return guarded([&]() -> int {
    if (!dp) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    const double n = 6.0 * dp->max_ncam;
    const double dense = n * n * n / 3.0;
    double env = dense;
    if (dp->has_big && !dp->env_first.empty()) {
      // tile products (128^3 multiply-adds each) of a factorisation that stays inside the envelope: an off-diagonal
      // tile of row i takes one per tile between the row's first column and its own, plus half a product for its
      // triangular solve; the diagonal tile half of that (symmetric) plus a sixth for its own factorisation
      const int nt = (int)dp->env_first.size() - 1;  // the last entry belongs to the right-hand side's tile row
      double prod = 0.0;
      for (int i = 0; i < nt; ++i) {
        const double w = i - dp->env_first[i] + 1;
        prod += (w - 1) * (w - 2) / 2 + (w - 1) / 2 + (w - 1) * 0.5 + 1.0 / 6.0;
      }
      env = std::min(dense, prod * 2.0 * kCholNB * kCholNB * kCholNB);
    }
    if (flops_dense) *flops_dense = dense;
    if (flops_envelope) *flops_envelope = env;
    return HIPBA_OK;
  });
}

int hipba_time_kernel(hipba_handle h, hipba_device_problem dp, const hipba_options* options, int32_t kernel,
                      int32_t repeats, double* avg_ms) {
  return guarded([&]() -> int {
    if (!h || !dp || !avg_ms || repeats <= 0) return fail(HIPBA_ERR_INVALID_ARGUMENT, "null argument");
    hipba_options o;
    if (options) o = *options; else hipba_options_default(&o);
    HIPBA_HIP(hipSetDevice(h->device));
    DevP P = dp->P;
    hipStream_t s = h->stream;
    LmParams lm = make_lm(o);
    if (lm.max_num_iterations < 1) lm.max_num_iterations = 1;
    const int n_act_blk = P.n_blk_pm + P.n_blk_vc;
    const int max_n = 6 * dp->max_ncam;
    const bool small = max_n <= kSmallMaxN;
    const int small_ncam = small ? dp->max_ncam : 0;
    const int schur_parts = schur_parts_for(P.n_windows, dp->max_pm_blocks);
    const bool direct = P.n_chunks > 0 && !small && o.linear_solver != HIPBA_LINEAR_PCG;  // as in a solve
    P.dense_S = direct ? dp->big.A : nullptr;
    P.dense_ld = direct ? dp->big.ld : 0;
    // bring the work arrays to the state of a first iteration (parameters are not modified)
    auto prepare = [&]() -> int {
      HIPBA_HIP(launch_cost(P, lm, 0, n_act_blk, 0, 1, false, s));
      HIPBA_HIP(launch_cost(P, lm, n_act_blk, P.n_blk_cc, 0, 1, false, s));
      HIPBA_HIP(launch_window_init(P, lm, kFused, 0, s));
      HIPBA_HIP(launch_camera_blocks(P, lm, 0, 0, s));
      HIPBA_HIP(launch_make_scale_c(P, lm.jacobi_scaling, s));
      HIPBA_HIP(launch_point_colnorm(P, lm, s));
      if (direct) HIPBA_HIP(big_zero_lower(P, dp->big, !dp->env_first.empty(), s));
      else if (dp->S_doubles) HIPBA_HIP(hipMemsetAsync(P.S, 0, sizeof(double) * (size_t)dp->S_doubles, s));
      HIPBA_HIP(launch_init_reduced(P, lm, 1, s));
      HIPBA_HIP(launch_point_schur(P, lm, small_ncam, schur_parts, s));
      return HIPBA_OK;
    };
    int solves_done = 0;
    auto solve_once = [&]() -> int {
      // (as in a solve: the first launch writes the table of tile sources, the others read it)
      if (small) HIPBA_HIP(launch_window_solve_small(P, lm, max_n, 0, solves_done++ == 0 ? 1 : 2, s));
      else if (o.linear_solver == HIPBA_LINEAR_PCG) HIPBA_HIP(big_solve_pcg(P, dp->big, o.pcg_max_iterations, o.pcg_tolerance, s));
      else HIPBA_HIP(big_solve_cholesky(P, dp->big, dp->env_first.empty() ? nullptr : dp->env_first.data(), s));
      return HIPBA_OK;
    };
    int rc = prepare();
    if (rc) return rc;
    if (kernel == HIPBA_KERNEL_BACKSUB) { rc = solve_once(); if (rc) return rc; }
    // the materialising kernel writes 2 + 12 + 6 doubles per observation into a scratch allocation
    double* jbuf = nullptr;
    const long long nj = P.n_act_obs;
    if (kernel == HIPBA_KERNEL_RESIDUAL_JACOBIAN) {
      if (P.n_windows != 1) return fail(HIPBA_ERR_UNSUPPORTED, "the materialising kernel is timed on single-window problems");
      HIPBA_HIP(hipMalloc((void**)&jbuf, sizeof(double) * 20 * (size_t)std::max<long long>(nj, 1)));
    }
    WinState st0;
    HIPBA_HIP(hipMemcpy(&st0, P.ws, sizeof(WinState), hipMemcpyDeviceToHost));
    WinDesc wd0;
    HIPBA_HIP(hipMemcpy(&wd0, P.wd, sizeof(WinDesc), hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    HIPBA_HIP(hipEventCreate(&e0));
    HIPBA_HIP(hipEventCreate(&e1));
    float total_ms = 0.f;
    auto one = [&]() -> int {
      switch (kernel) {
        case HIPBA_KERNEL_RESIDUAL: HIPBA_HIP(launch_cost(P, lm, 0, n_act_blk, 0, 1, true, s)); break;
        case HIPBA_KERNEL_RESIDUAL_JACOBIAN:
          HIPBA_HIP(launch_residual_jacobian(wd0.K, lm, 1, nj, P.n_poses, P.n_points, P.poses[st0.sel], P.points[st0.sel], P.obs_pose, P.obs_dp,
                                             P.obs_uv, jbuf, jbuf + 2 * nj, jbuf + 14 * nj, nullptr, nullptr, s));
          break;
        case HIPBA_KERNEL_CAMERA_BLOCKS: HIPBA_HIP(launch_camera_blocks(P, lm, 1, 0, s)); break;
        case HIPBA_KERNEL_POINT_SCHUR: HIPBA_HIP(launch_point_schur(P, lm, small_ncam, schur_parts, s)); break;
        case HIPBA_KERNEL_BACKSUB: HIPBA_HIP(launch_backsub(P, lm, backsub_can_carry_cost(P), s)); break;
        case HIPBA_KERNEL_REDUCED_SOLVE: return solve_once();
        default: return fail(HIPBA_ERR_INVALID_ARGUMENT, "kernel %d cannot be timed on a resident problem", kernel);
      }
      return HIPBA_OK;
    };
    rc = one();  // warm-up
    if (!rc) {
      if (kernel == HIPBA_KERNEL_REDUCED_SOLVE) {
        // the factorisation destroys nothing it needs again, but the Schur matrix must be rebuilt per repeat
        for (int i = 0; i < repeats && !rc; ++i) {
          if (direct) HIPBA_HIP(big_zero_lower(P, dp->big, !dp->env_first.empty(), s));
          else if (dp->S_doubles) HIPBA_HIP(hipMemsetAsync(P.S, 0, sizeof(double) * (size_t)dp->S_doubles, s));
          HIPBA_HIP(launch_init_reduced(P, lm, 1, s));
          HIPBA_HIP(launch_point_schur(P, lm, small_ncam, schur_parts, s));
          HIPBA_HIP(hipEventRecord(e0, s));
          rc = one();
          HIPBA_HIP(hipEventRecord(e1, s));
          HIPBA_HIP(hipEventSynchronize(e1));
          float ms = 0.f;
          HIPBA_HIP(hipEventElapsedTime(&ms, e0, e1));
          total_ms += ms;
        }
      } else {
        HIPBA_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < repeats && !rc; ++i) rc = one();
        HIPBA_HIP(hipEventRecord(e1, s));
        HIPBA_HIP(hipEventSynchronize(e1));
        HIPBA_HIP(hipEventElapsedTime(&total_ms, e0, e1));
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (jbuf) { (void)hipStreamSynchronize(s); (void)hipFree(jbuf); }
    if (rc) return rc;
    *avg_ms = (double)total_ms / repeats;
    // leave the window state as a finished solve would: nothing active
    HIPBA_HIP(hipStreamSynchronize(s));
    return HIPBA_OK;
  });
}

}  // extern "C"
