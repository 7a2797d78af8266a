// Internal definitions shared by the gfx950 kernels and the C-ABI glue of librls_mi355x.so.
// Public contract: include/rls_mi355x.h.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/rls_mi355x.h"

#define RLS_WAVE 64

struct rls_tuning {
  int gemvn_g = 0;      // lanes per row group in gemv_n (0 = heuristic)
  int gemvn_waves = 0;  // waves per workgroup in gemv_n (0 = heuristic)
  int gemvt_cols = 0;   // columns per workgroup in gemv_t (0 = heuristic)
  int gemvt_reverse = -1;  // gemv_t walks the columns from the LAST one down (1), from the first up (0), or (-1, default) from the
                           // last down exactly when A is larger than the Infinity Cache: the second product of the two-GEMV
                           // normal operator then starts on the columns the first one has just left in the cache
  int graph_chunk = 16; // iterations captured per hipGraph
  int use_graph = 1;
  int fuse_level = 1;   // 0: separate BLAS-1 style update kernel; 1: fused update
  int fused_normal = 1; // 1: one-pass register-slab normal operator when the shape allows it
  int cgnr_pipeline = 1; // 1: CGNR as slab kernel (with the CG update in its prologue) + reduce kernel
  int batched_mfma = 1;  // 1: batched plans run the two skinny products on the matrix cores (skinny.hip)
  int gram_pipeline = 1; // 1: Gram-mode CGNR as one launch per iteration (normal.hip)
  int pipe_hint_mode = 0; // (r, p) pair hints of the slab pipeline: 0 = host bookkeeping, 1 = always "unknown",
                          // 2 = deliberately wrong (tests: exercises the kernel's check-and-reload path)
  int resident = 1;       // 1: single-RHS matrix-free CGNR whose A fits the register files runs a whole step call as
                          // ONE launch (normal.hip, cgnr_resident_kernel)
  int resident_preclear = 1; // 1: the init kernels zero the resident kernels' arrival counters (no memset launch ahead of the first step)
  int resident_server = 1;     // 1: rls_cgnr_step_status leaves the resident kernel listening for the next call (rls_cg_start::srv_ctl)
  int resident_server_idle_us = 300;  // ... for this long
  int resident_ahead = 1;      // 1: a listening kernel computes one iteration ahead of the next command (the SPEC instantiations, normal.hip)
  int fista_defer = 1;         // 1: fista_resident_kernel sums ||res||^2 off the critical path where it can (normal.hip, DEFER); 0: measurement
  int resident_l2_rows = 1;    // 1: the matrix-free resident kernels keep their partial rows in the XCD's L2 when the placement allows (normal.hip,
                               // resident_rows_at_l2); 0: always written through (measurement switch)
  int small = 1;               // 1: systems that fit ONE CU's registers run a whole step call as a single-workgroup launch (small.hip)
  int status_mailbox = 2;      // >= 1: status read-backs are a kernel writing into pinned host memory + a host spin on its
                               // sequence word (rls_fetch_*); 2: and rls_*_step_status has the call's LAST kernel do that
                               // store where it can (rls_mailbox_slot); 0: hipMemcpyAsync + stream wait
  // ---- per-kernel-family measurement switches (rounds 1-5 kept these as file-scope statics; a context is the unit of
  //      re-entrancy -- SURVEY 8b: all mutable state in rls_ctx -- so they live here, one copy per context) ----
  int slab_g = 0;              // normal.hip: force the lanes per row chunk of the register slab (0 = heuristic); set before the operator is created
  int slab_order = 1;          // 0: wait for the small loads before the slab goes out, 1: barrier only
  int resident_barrier = 2;    // matrix-free resident kernels' exchange: 2 = two-level where the grid allows, 1 = flat
  int red_threads = 1024;      // reduce kernels: 16 columns x 64 row groups per workgroup
  int slab_multi = 1;          // shapes with more row blocks than CUs: one workgroup walks several blocks
  int skinny_t_waves = 4, skinny_t_u = 4, skinny_v_waves = 4, skinny_v_u = 1, skinny_v_splits = 0;  // skinny.hip (tools/skinny_probe.py)
  int skinny_t_roll = 8, skinny_v_roll = 2, skinny_g_roll = 8;  // rolling-window depth of the complex T / V / Gram products (0 = two-set pipeline)
  int skinny_half = 1;         // the (re | im) operand packing for <= 8 complex right-hand sides
  int gram_lds = 96 * 1024;    // dynamic LDS requested by the Gram tile kernel as an occupancy limiter
  int64_t tv_fused_max_n = 2048;  // tv.hip: larger images run 2 chip-wide launches per FGP iteration instead of one CU
  int tv_fused_2d = 1;         // the register-resident 2-D FGP kernel (n <= 8192 pixels)
  int kaczmarz_nt = 0;         // kaczmarz.hip: workgroup size override (0 = heuristic)
  int resident_spin = 100000;  // bound of every in-kernel wait, in polls (~1 us each: a wall-clock bound of ~0.1 s per
                               // wait); a launch that runs into it is a no-op and the host re-runs its iterations on the
                               // per-iteration pipeline (solvers.hip, *_recover)
};

constexpr int RLS_FETCH_MAX = 24;  // blocks one publishing kernel moves (the statuses of a group of small problems: rls_cgnr_get_status_group)
struct rls_ctx {
  int device = 0;
  void* server = nullptr;  // the plan whose resident kernel is alive in server mode on this context's stream (rls_server_stop)
  uint64_t id = 0;  // unique per context ever created in this process (a plan that outlives its context compares it: rls_ctx_alive)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  char err[512] = {0};
  // reduction scratch: partial sums (double) + a few result slots, and a pinned host mirror
  double* red_d = nullptr;  // [RLS_RED_SLOTS] doubles
  float* res_d = nullptr;   // small float result block on device
  float* res_h = nullptr;   // pinned host mirror
  rls_tuning tune;
  uint64_t tune_epoch = 0;    // bumped by every rls_tune_set: a cached hipGraph captured under other settings is dropped (solvers.hip, run_steps)
  int cus = 0;                // compute units of `device` (looked up once, by whoever asks first on this context)
  int resident_failures = 0;  // resident launches of this context that timed out; at 2 the context stops using them
  bool pools = false;         // device memory comes from the device's stream-ordered pool (rls_dev_alloc)
  // status mailbox (rls_fetch_add / rls_fetch_wait): device -> pinned copies queued for ONE publishing launch
  struct fetch_item {
    const void* src;
    void* dst;
    unsigned dwords;
  } fq[RLS_FETCH_MAX];
  int nfq = 0;
  unsigned* mb_h = nullptr;  // pinned, host-mapped sequence word the publishing kernel stores last (system scope)
  unsigned mb_seq = 0;
};

// Status read-backs without a D2H copy engine in the way: the reference's solve! loop evaluates done() and fires callbacks
// after EVERY iterate (src/RegularizedLeastSquares.jl:103-117), so one read-back per iteration is on the critical path of a
// solve with callbacks.  rls_fetch_add queues "copy `bytes` (a multiple of 4) from device memory to this PINNED host block";
// rls_fetch_wait enqueues ONE small kernel that stores everything queued straight into the host blocks and then a sequence
// word, and spins on that word (bounded; falls back to a stream wait).  With tune.status_mailbox = 0: hipMemcpyAsync + wait.
int32_t rls_fetch_add(rls_ctx* ctx, const void* src_d, void* dst_pinned, size_t bytes);
int32_t rls_fetch_wait(rls_ctx* ctx);
// The last kernel of a step call can publish the plan's scalars itself (one launch and one launch boundary less on the per-iterate
// path): the host arms a slot -- where to store (the plan's pinned mirror), the context's sequence word and the value to put there --,
// hands it to that kernel and waits for the sequence value.  dst == nullptr: not armed.
struct rls_mailbox_slot {
  void* dst = nullptr;
  unsigned* seq_h = nullptr;
  unsigned seq = 0;
};
rls_mailbox_slot rls_mailbox_arm(rls_ctx* ctx, void* dst_pinned);
// server mode of the resident kernels (see rls_cg_start::srv_ctl for the protocol): what a launch that is to stay and LISTEN gets
constexpr unsigned RLS_SRV_EXIT = 0xffffffffu;
constexpr unsigned RLS_SRV_AHEAD = 0x80000000u;  // bit 31 of rls_srv_args::idle_us, read by the single-workgroup kernels (small.hip): one iteration ahead
struct rls_srv_args {
  unsigned* ctl = nullptr;  // control block in pinned host memory (null: an ordinary launch)
  unsigned seq0 = 0, idle_us = 0;
  rls_mailbox_slot mb;
};
int32_t rls_mailbox_wait(rls_ctx* ctx, unsigned seq);
#ifdef __HIPCC__
// Orders this wave's system-scope (write-through, sc0 sc1) stores ahead of its next one: their acknowledgements are back.  A release
// FENCE at system scope would also write back every dirty line of this XCD's L2 (buffer_wbl2) -- the vectors the kernel has just
// written -- before the sequence word may go out: 2.5 us per status on the critical path for data the host does not read
// (tools/bench_cadence.py: 27.0 -> 24.3 us per call at 256 x 128).  The mailbox payload itself never sits in L2 dirty: every dword of
// it is a system-scope atomic store to fine-grained host memory.
__device__ static inline void rls_system_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Called by ALL 64 lanes of ONE wave of the call's last kernel, behind its last update of the struct, every lane holding the same
// v: lane i stores dword i into pinned host memory (system scope; ONE store instruction -- twenty single-lane stores are twenty
// PCIe writes and cost 2.3 us more per status), then lane 0 -- behind their acknowledgements -- the sequence word.
template <typename S>
__device__ static inline void rls_mailbox_publish(const rls_mailbox_slot& mb, const S& v, int lane) {
  constexpr unsigned ND = sizeof(S) / 4;
  static_assert(sizeof(S) % 4 == 0 && ND <= 64, "status structs are published dword by dword by one wave");
  if (!mb.dst) return;
  const unsigned* src = reinterpret_cast<const unsigned*>(&v);
  unsigned mine = 0;
#pragma unroll
  for (unsigned i = 0; i < ND; ++i) mine = lane == (int)i ? src[i] : mine;
  if (lane < (int)ND) __hip_atomic_store(reinterpret_cast<unsigned*>(mb.dst) + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  rls_system_stores_done();
  if (lane == 0) __hip_atomic_store(mb.seq_h, mb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

// ---- memory ------------------------------------------------------------------------------------------------------------
// Device memory of the library (plan scratch, rls_malloc) is STREAM-ORDERED on the context's stream: hipMallocAsync /
// hipFreeAsync on the device's default memory pool, whose release threshold is raised once so that freed blocks stay
// cached.  A whole solve! creates and drops a dozen scratch vectors; with hipMalloc / hipFree that was a dozen driver
// calls plus a device-wide synchronisation per free (hipFree waits for EVERY stream of the device -- eight solver threads
// serialised each other through it: bench.py cgnr_distinct_A_8_problems, 8 streams slower than 1).  Reuse is ordered by
// the stream, as with any stream-ordered allocator: memory handed to ANOTHER context's stream must be synchronised by the
// caller before it is freed.  Devices without memory pools (or RLS_ALLOC=sync) fall back to hipMalloc / hipFree.
// Small pinned host blocks (the status mirrors of the plans) come from a process-wide free list: hipHostMalloc costs
// ~100 us a call.
hipError_t rls_dev_alloc(rls_ctx* ctx, void** p, size_t bytes);
hipError_t rls_dev_free(rls_ctx* ctx, void* p);   // ctx may be null (or destroyed: pass null): synchronous hipFree
hipError_t rls_pinned_alloc(void** p, size_t bytes);
void rls_pinned_free(void* p);
// Every entry point that is about to put work on the context's stream (or to tear something down) comes through here: a resident
// kernel left listening in server mode (rls_cgnr_step_status) is asked to leave first -- anything queued behind it would otherwise
// wait for its idle timeout.
void rls_server_stop(rls_ctx* ctx);
static inline hipError_t rls_enter(rls_ctx* ctx) {
  if (ctx->server) rls_server_stop(ctx);
  return hipSetDevice(ctx->device);
}
bool rls_ctx_alive(const rls_ctx* ctx, uint64_t id);  // this very context (same address AND same generation id) still exists
// The allocation calls of a plan's create / destroy function go to the context named by the innermost live scope on
// this thread (null: the synchronous calls).
struct rls_alloc_scope {
  rls_ctx* prev;
  explicit rls_alloc_scope(rls_ctx* ctx);
  ~rls_alloc_scope();
};
hipError_t rls_scoped_malloc(void** p, size_t bytes);
hipError_t rls_scoped_free(void* p);

// Host-side waits poll (hipStreamQuery / hipEventQuery) instead of blocking.  A blocking wait sleeps on an interrupt and
// on this stack now and then wakes up tens of milliseconds late (tools/stall_probe2.py: wall 123 ms for 75 ms of
// events) -- invisible in hipEvent times, but it is wall-clock latency of every status read-back and of the solve
// as a whole.  The spin is bounded by WALL CLOCK (250 ms: status read-backs and whole solves of the BASELINE configs
// finish inside it); longer waits hand the core back and block, so a long enqueue cannot pin a host core for minutes.
static inline void rls_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield");
#endif
}
template <typename Q, typename B>
static inline hipError_t rls_bounded_spin(Q&& query, B&& block) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 0;; ++n) {
    const hipError_t e = query();
    if (e != hipErrorNotReady) return e;
    rls_cpu_relax();
    if ((n & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(250)) break;
  }
  return block();
}
static inline hipError_t rls_stream_wait(hipStream_t s) {
  return rls_bounded_spin([s] { return hipStreamQuery(s); }, [s] { return hipStreamSynchronize(s); });
}
static inline hipError_t rls_event_wait(hipEvent_t ev) {
  return rls_bounded_spin([ev] { return hipEventQuery(ev); }, [ev] { return hipEventSynchronize(ev); });
}

// One process-wide lock around (a) every hipGraph capture of the library (begin ... end capture: host-side recording
// of a few dozen launches) and (b) the cross-stream event chain of the resident kernels.  While ANY stream of the
// process is capturing, hipStreamWaitEvent from another thread on an event recorded outside that capture can fail
// with hipErrorStreamCaptureIsolation (seen with one context capturing a FISTA graph while another chained a
// resident launch: 1 run in 6); both sections are short and host-only, so serialising them costs nothing measurable.
void rls_resident_forget(int device, hipStream_t stream);  // solvers.hip: a stream is going away (resident launch chain)
inline std::mutex& rls_capture_mutex() {
  static std::mutex m;
  return m;
}

// "done once per device" guard for per-function attributes (hipFuncSetAttribute applies to the function on the CURRENT
// device: a process that drives several GPUs, rls_comm_*, must set it on each of them); race-free across host threads
#include <atomic>
// Per-device one-time work (the kernels' > 64 KiB dynamic-LDS attributes) that nobody may run past before it is COMPLETE: two host
// threads entering their first solves at the same time (src/MultiThreading.jl:71) used to race here -- the second saw "seen" while
// the first was still setting attributes and launched a kernel whose LDS size was not allowed yet (a cold-process failure of
// tests/test_gpu_contexts.py, one run in a few).  Usage: `if (auto once = attr_once.first(device)) { ...work... }` -- the guard
// lives through the body, publishes "done" behind it and only then lets the waiting threads go.
struct rls_device_once {
  std::atomic<uint64_t> done{0};
  std::mutex mu;
  struct guard {
    rls_device_once* o;
    uint64_t bit;
    bool run;
    guard(rls_device_once* o_, uint64_t bit_, bool run_) : o(o_), bit(bit_), run(run_) {}
    guard(const guard&) = delete;
    guard(guard&& g) : o(g.o), bit(g.bit), run(g.run) { g.run = false; }
    ~guard() {
      if (run) {
        o->done.fetch_or(bit, std::memory_order_release);
        o->mu.unlock();
      }
    }
    explicit operator bool() const { return run; }
  };
  guard first(int device) {  // true for exactly one caller per device; the others return once that caller's body has finished
    const uint64_t bit = 1ull << (device & 63);
    if (done.load(std::memory_order_acquire) & bit) return guard(this, bit, false);
    mu.lock();
    if (done.load(std::memory_order_acquire) & bit) {
      mu.unlock();
      return guard(this, bit, false);
    }
    return guard(this, bit, true);
  }
};

constexpr int RLS_RED_SLOTS = 4096;
constexpr int RLS_RES_FLOATS = 64;

#define RLS_CHECK_CTX(ctx)            \
  do {                                \
    if (!(ctx)) return RLS_E_INVALID; \
  } while (0)

static inline int32_t rls_fail(rls_ctx* ctx, int32_t code, const char* what) {
  if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s (code %d)", what, (int)code);
  return code;
}

#define RLS_HIP(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      if (ctx)                                                                                   \
        snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #expr,                 \
                 hipGetErrorString(_e), __FILE__, __LINE__);                                     \
      return (int32_t)_e;                                                                        \
    }                                                                                            \
  } while (0)

#define RLS_TRY(expr)            \
  do {                           \
    int32_t _s = (expr);         \
    if (_s != 0) return _s;      \
  } while (0)

static inline size_t rls_elem_size(int32_t dtype) { return dtype == RLS_C32 ? 8 : 4; }
static inline bool rls_dtype_ok(int32_t dtype) { return dtype == RLS_F32 || dtype == RLS_C32; }

// ---------------------------------------------------------------------------------------------
// device-side scalar helpers: element type E is float (RLS_F32) or float2 (RLS_C32)
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
template <typename E>
struct elem;
template <>
struct elem<float> {
  static constexpr bool cplx = false;
  static constexpr int vec = 4;  // elements per 16-byte load
  __device__ static inline float zero() { return 0.f; }
  __device__ static inline float make(float re, float) { return re; }
  __device__ static inline float mul(float a, float b) { return a * b; }
  __device__ static inline float mulc(float a, float b) { return a * b; }  // conj(a)*b
  __device__ static inline float fma(float a, float b, float c) { return fmaf(a, b, c); }
  __device__ static inline float fmac(float a, float b, float c) { return fmaf(a, b, c); }  // conj(a)*b + c
  __device__ static inline float fma_pk(float a, float b, float c) { return fmaf(a, b, c); }
  __device__ static inline float fmac_pk(float a, float b, float c) { return fmaf(a, b, c); }
  __device__ static inline float add(float a, float b) { return a + b; }
  __device__ static inline float sub(float a, float b) { return a - b; }
  __device__ static inline float scale(float s, float a) { return s * a; }
  __device__ static inline float abs2(float a) { return a * a; }
  __device__ static inline float absv(float a) { return fabsf(a); }
  __device__ static inline float re(float a) { return a; }
  __device__ static inline float im(float) { return 0.f; }
};
template <>
struct elem<float2> {
  static constexpr bool cplx = true;
  static constexpr int vec = 2;
  __device__ static inline float2 zero() { return make_float2(0.f, 0.f); }
  __device__ static inline float2 make(float re, float im) { return make_float2(re, im); }
  __device__ static inline float2 mul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
  }
  __device__ static inline float2 mulc(float2 a, float2 b) {  // conj(a) * b
    return make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
  }
  __device__ static inline float2 fma(float2 a, float2 b, float2 c) {  // a*b + c
    float re = fmaf(a.x, b.x, c.x);
    float im = fmaf(a.x, b.y, c.y);
    re = fmaf(-a.y, b.y, re);
    im = fmaf(a.y, b.x, im);
    return make_float2(re, im);
  }
  __device__ static inline float2 fmac(float2 a, float2 b, float2 c) {  // conj(a)*b + c
    float re = fmaf(a.x, b.x, c.x);
    float im = fmaf(a.x, b.y, c.y);
    re = fmaf(a.y, b.y, re);
    im = fmaf(-a.y, b.x, im);
    return make_float2(re, im);
  }
  // The same two operations as TWO packed FMAs (v_pk_fma_f32, full rate on CDNA3/4) instead of four scalar ones;
  // op_sel / op_sel_hi pick the halves, neg_lo / neg_hi the sign, so no operand is copied or swizzled and the
  // four roundings are the ones of fma() / fmac() above, in the same order (bit-identical results).  For the
  // VALU-bound product loops over the register slab; hipcc's own SLP packing of these loops needed register
  // copies of the slab and spilled (Makefile: -fno-slp-vectorize).
  typedef float v2f __attribute__((ext_vector_type(2)));
  __device__ static inline float2 fma_pk(float2 a, float2 b, float2 c) {  // a*b + c
    v2f A = {a.x, a.y}, B = {b.x, b.y}, C = {c.x, c.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(C) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(C) : "v"(A), "v"(B));
    return make_float2(C.x, C.y);
  }
  __device__ static inline float2 fmac_pk(float2 a, float2 b, float2 c) {  // conj(a)*b + c
    v2f A = {a.x, a.y}, B = {b.x, b.y}, C = {c.x, c.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(C) : "v"(A), "v"(B));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(C) : "v"(A), "v"(B));
    return make_float2(C.x, C.y);
  }
  __device__ static inline float2 add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
  __device__ static inline float2 sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
  __device__ static inline float2 scale(float s, float2 a) { return make_float2(s * a.x, s * a.y); }
  __device__ static inline float abs2(float2 a) { return fmaf(a.x, a.x, a.y * a.y); }
  __device__ static inline float absv(float2 a) { return hypotf(a.x, a.y); }
  __device__ static inline float re(float2 a) { return a.x; }
  __device__ static inline float im(float2 a) { return a.y; }
};

// Float32 multiply / add / subtract that the compiler cannot fuse into an FMA.  The library is built with
// -ffp-contract=fast, under which the backend fuses a*b + c whatever the source says (`#pragma clang fp contract(off)`
// and the *_rn intrinsics, which are plain operators to clang, do not stop it).  Where a device scalar must come
// out bit-identical to the host's Float32 arithmetic (NumPy scalars: one rounding per operation) -- stopping
// tests, the POGM coefficient recurrences -- the operations go through single instructions.
__device__ static inline float f32_mul(float a, float b) {
  float r;
  asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ static inline float f32_add(float a, float b) {
  float r;
  asm volatile("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ static inline float f32_sub(float a, float b) {
  float r;
  asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// complex scalar carried in double precision on the device (alpha, beta, dot results)
struct dcomplex {
  double re, im;
};
__device__ static inline dcomplex dc_div(dcomplex a, dcomplex b) {
  double d = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}

// wave-level sum (64 lanes), result valid in every lane.  The first four butterfly steps stay
// inside a row of 16 lanes and use DPP (full-rate VALU): quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror; only the two cross-row steps go through ds_bpermute.
__device__ static inline float dpp_f(float v, int ctrl) {
  switch (ctrl) {
    case 0xB1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    case 0x4E: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    case 0x141: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    case 0x140: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  }
}
__device__ static inline double dpp_d(double v, int ctrl) {
  const long long b = __builtin_bit_cast(long long, v);
  int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  lo = __builtin_bit_cast(int, dpp_f(__builtin_bit_cast(float, lo), ctrl));
  hi = __builtin_bit_cast(int, dpp_f(__builtin_bit_cast(float, hi), ctrl));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// The two steps ACROSS rows: gfx950's v_permlane16_swap / v_permlane32_swap (VALU; the odd rows of the first operand change places
// with the even rows of the second, the upper half of the wave with the lower half) instead of ds_bpermute -- with both operands
// the same value, a' + b' is v[l] + v[l ^ 16] (v[l] + v[l ^ 32]) in every lane: the butterfly's pairs, the same bits (addition
// commutes), without an LDS round trip per step.  Inline assembly with its own wait states: hipcc 7.2's builtin hands back the first
// result's register for BOTH results, and an asm statement is invisible to the hazard recogniser (tools/ubench/permlane_probe.hip).
// (returns a' + b': the sum of the pair, in both of its lanes)
__device__ static inline float pair_sum16(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ static inline float pair_sum32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ static inline double pair_sum16(double v) {
  const long long q = __builtin_bit_cast(long long, v);
  float al = __builtin_bit_cast(float, (int)(q & 0xffffffffll)), ah = __builtin_bit_cast(float, (int)(q >> 32));
  float bl = al, bh = ah;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1" : "+v"(al), "+v"(bl), "+v"(ah), "+v"(bh));
  const double a = __builtin_bit_cast(double, ((long long)__builtin_bit_cast(int, ah) << 32) | (long long)(unsigned)__builtin_bit_cast(int, al));
  const double b = __builtin_bit_cast(double, ((long long)__builtin_bit_cast(int, bh) << 32) | (long long)(unsigned)__builtin_bit_cast(int, bl));
  return a + b;
}
__device__ static inline double pair_sum32(double v) {
  const long long q = __builtin_bit_cast(long long, v);
  float al = __builtin_bit_cast(float, (int)(q & 0xffffffffll)), ah = __builtin_bit_cast(float, (int)(q >> 32));
  float bl = al, bh = ah;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(al), "+v"(bl), "+v"(ah), "+v"(bh));
  const double a = __builtin_bit_cast(double, ((long long)__builtin_bit_cast(int, ah) << 32) | (long long)(unsigned)__builtin_bit_cast(int, al));
  const double b = __builtin_bit_cast(double, ((long long)__builtin_bit_cast(int, bh) << 32) | (long long)(unsigned)__builtin_bit_cast(int, bl));
  return a + b;
}
__device__ static inline double wave_sum(double v) {
  v += dpp_d(v, 0xB1);
  v += dpp_d(v, 0x4E);
  v += dpp_d(v, 0x141);
  v += dpp_d(v, 0x140);
  v = pair_sum16(v);
  v = pair_sum32(v);
  return v;
}
__device__ static inline float wave_sum(float v) {
  v += dpp_f(v, 0xB1);
  v += dpp_f(v, 0x4E);
  v += dpp_f(v, 0x141);
  v += dpp_f(v, 0x140);
  v = pair_sum16(v);
  v = pair_sum32(v);
  return v;
}

// v + (the value of lane ^ off): the butterfly step of the in-wave sums of the slab kernels.  `off` is a constant after unrolling:
// 16 and 32 are the permlane swaps above, 8 stays inside a row (row_ror:8 is the same exchange), the rest go through ds_bpermute.
__device__ static __forceinline__ float add_xor(float v, int off) {
  if (off == 32) return pair_sum32(v);
  if (off == 16) return pair_sum16(v);
  if (off == 8) return v + dpp_f(v, 0x128);
  return v + __shfl_xor(v, off, 64);
}

// block-level sum of up to 16 waves; `smem` needs 16 doubles; result valid in every thread.
// Deterministic: fixed tree inside the wave, fixed order across waves.
__device__ static inline double block_sum(double v, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();  // protect smem reuse across consecutive calls
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < nw; ++i) s += smem[i];
  return s;
}

// The same with the wave count as a compile-time constant.  blockDim.x is read from the AQL dispatch packet
// (s_load through the dispatch pointer); the packet lives in the queue ring in host-visible memory, so that
// read costs microseconds and depends on the XCD -- harmless where it hides under other waits, 15 us per
// launch where a barrier or an lgkmcnt(0) wait sits right behind it (measured on the Gram-mode FISTA kernel).
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding vector-memory
// load (s_waitcnt vmcnt(0)); in the one-pass kernels that is the whole register slab of A, i.e. the reductions of
// the CG / FISTA update would sit out the slab's flight instead of running under it.
__device__ static inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NW>
__device__ static inline double block_sum_n(double v, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_sum(v);
  lds_barrier();
  if (lane == 0) smem[w] = v;
  lds_barrier();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += smem[i];
  return s;
}
template <int NW>
__device__ static inline void block_sum3_n(double& a, double& b, double& c, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  lds_barrier();
  if (lane == 0) {
    smem[w] = a;
    smem[16 + w] = b;
    smem[32 + w] = c;
  }
  lds_barrier();
  double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    sa += smem[i];
    sb += smem[16 + i];
    sc += smem[32 + i];
  }
  a = sa;
  b = sb;
  c = sc;
}

// The same two reductions WITHOUT their leading barrier, for callers that alternate between two disjoint scratch areas and
// have a workgroup barrier between any read of an area and the next write to it (the resident kernels: sum3 in slots
// [0,8) [16,24) [32,40) of `smem`, the single sum in slots [8,16) -- each is rewritten only after the OTHER one's barrier).
template <int NW>
__device__ static inline void block_sum3_nolead(double& a, double& b, double& c, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  if (lane == 0) {
    smem[w] = a;
    smem[16 + w] = b;
    smem[32 + w] = c;
  }
  lds_barrier();
  double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    sa += smem[i];
    sb += smem[16 + i];
    sc += smem[32 + i];
  }
  a = sa;
  b = sb;
  c = sc;
}
template <int NW, int BASE = 8>
__device__ static inline double block_sum_nolead(double v, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) smem[BASE + w] = v;
  lds_barrier();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += smem[BASE + i];
  return s;
}
// sum over lanes 0..31 of a wave (row pairs of 16: four DPP steps inside a row, one shuffle across): complete in lanes 0..15 and
// 16..31 alike; lanes >= 32 hold the sum of THEIR half.  Fixed order.
__device__ static inline double half_wave_sum(double v) {
  v += dpp_d(v, 0xB1);
  v += dpp_d(v, 0x4E);
  v += dpp_d(v, 0x141);
  v += dpp_d(v, 0x140);
  v = pair_sum16(v);
  return v;
}

// three sums with one barrier pair; `smem` needs 48 doubles
__device__ static inline void block_sum3(double& a, double& b, double& c, double* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  a = wave_sum(a);
  b = wave_sum(b);
  c = wave_sum(c);
  __syncthreads();
  if (lane == 0) {
    smem[w] = a;
    smem[16 + w] = b;
    smem[32 + w] = c;
  }
  __syncthreads();
  double sa = 0.0, sb = 0.0, sc = 0.0;
  for (int i = 0; i < nw; ++i) {
    sa += smem[i];
    sb += smem[16 + i];
    sc += smem[32 + i];
  }
  a = sa;
  b = sb;
  c = sc;
}

// elementwise prox / projection used by the fused FISTA updates (src/proximalMaps/ProxL1.jl:18-22,
// ProxL2.jl:18-21, src/Utils.jl:114-144); kinds are the RLS_REG_* / RLS_PROJ_* codes of the C ABI
template <typename E>
__device__ static inline E fista_prox_elem(E v, int reg_kind, float thr) {
  if (reg_kind == RLS_REG_L1) {
    const float eps = 1.1920929e-07f;
    const float a = elem<E>::absv(v);
    const float sh = fmaxf(a - thr, 0.f);
    const float den = a + eps;
    return elem<E>::make((sh * (elem<E>::re(v) + eps)) / den, (sh * elem<E>::im(v)) / den);
  }
  if (reg_kind == RLS_REG_L2) {
    const double f = 1.0 / (1.0 + 2.0 * (double)thr);
    return elem<E>::make((float)((double)elem<E>::re(v) * f), (float)((double)elem<E>::im(v) * f));
  }
  return v;
}
template <typename E>
__device__ static inline E fista_proj_elem(E v, int proj_kind) {
  if (proj_kind == RLS_PROJ_NONE) return v;
  float re = elem<E>::re(v);
  if (proj_kind == RLS_PROJ_POSITIVE && re < 0.f) re = 0.f;
  return elem<E>::make(re, 0.f);
}

// 16-byte (or element-sized) register chunks of a matrix column
typedef float f4 __attribute__((ext_vector_type(4)));

template <typename E, int NV>
struct chunk {
  E e[NV];
};

template <typename E, int NV>
__device__ static inline chunk<E, NV> load_chunk(const E* p) {
  chunk<E, NV> c;
  if constexpr (NV * sizeof(E) == 16) {
    f4 v = *reinterpret_cast<const f4*>(p);
    c = __builtin_bit_cast(chunk<E, NV>, v);
  } else {
    static_assert(NV == 1, "scalar chunk");
    c.e[0] = *p;
  }
  return c;
}
template <typename E, int NV>
__device__ static inline chunk<E, NV> zero_chunk() {
  chunk<E, NV> c;
#pragma unroll
  for (int i = 0; i < NV; ++i) c.e[i] = elem<E>::zero();
  return c;
}

#endif  // __HIPCC__

// device-resident CGNR scalars (src/CGNR.jl:13-24) plus the state of the fused pipeline
struct cgnr_scalars {
  double rr;     // ||r||^2 now
  double z0;     // ||A^H b||
  double zeta;   // ||r||^2 at the start of the last iteration
  double alpha_re, alpha_im, beta_re, beta_im;
  float lambda, rel_tol;
  int iteration, max_iter, done;
  int pending;   // v and its partial dots are computed for the current p; the update is not applied yet
  int cur;       // which (r, p) pair is current: 0 = the caller's vectors, 1 = the plan's scratch
  int fresh;     // staging copy only: this round's K_A produced slab partials
};

// device-resident FISTA scalars (src/FISTA.jl:15-27) plus the state of the fused pipeline
struct fista_scalars {
  double norm_x0, res_norm, rel_res_norm;
  float rho, theta, theta_old, rel_tol, lambda;
  int iteration, max_iter, done, restart, reg_kind, proj_kind;
  long long l21_slices;
  int pending;  // res_raw = AHA y[ycur] is computed; the gradient/prox update is not applied yet
  int ycur;     // which extrapolated-point buffer is current
  int fresh;    // staging copy only
};
// Field-by-field copy.  A plain struct assignment lets the compiler move the fields nobody touches as one
// byte blob; when that blob is 12 bytes it stays in a private alloca, the alloca is promoted to LDS and
// indexed with the flat thread id, whose workgroup sizes are read from the AQL dispatch packet in
// host-visible memory: 15 us per launch on the Gram-mode FISTA kernel (measured, XCD-dependent).
#define RLS_FISTA_COPY(dst, src)        \
  do {                                  \
    (dst).norm_x0 = (src).norm_x0;      \
    (dst).res_norm = (src).res_norm;    \
    (dst).rel_res_norm = (src).rel_res_norm; \
    (dst).rho = (src).rho;              \
    (dst).theta = (src).theta;          \
    (dst).theta_old = (src).theta_old;  \
    (dst).rel_tol = (src).rel_tol;      \
    (dst).lambda = (src).lambda;        \
    (dst).iteration = (src).iteration;  \
    (dst).max_iter = (src).max_iter;    \
    (dst).done = (src).done;            \
    (dst).restart = (src).restart;      \
    (dst).reg_kind = (src).reg_kind;    \
    (dst).proj_kind = (src).proj_kind;  \
    (dst).l21_slices = (src).l21_slices; \
    (dst).pending = (src).pending;      \
    (dst).ycur = (src).ycur;            \
    (dst).fresh = (src).fresh;          \
  } while (0)

// Wave-uniform scalars into SGPRs.  The solver scalars that a resident kernel carries from one iteration to the next come
// out of block reductions (LDS reads: VGPRs), and the compiler does not know they are uniform: left alone they sit in ~20
// vector registers per lane for the whole loop, beside a 128-register slab.  One v_readfirstlane per dword per iteration
// moves them to the scalar file.
#ifdef __HIPCC__
__device__ static inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ static inline float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ static inline double uni(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ static inline long long uni(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (long long)lo;
}
#define RLS_CGNR_UNIFORM(S)                                                       \
  do {                                                                            \
    (S).rr = uni((S).rr); (S).z0 = uni((S).z0); (S).zeta = uni((S).zeta);         \
    (S).alpha_re = uni((S).alpha_re); (S).alpha_im = uni((S).alpha_im);           \
    (S).beta_re = uni((S).beta_re); (S).beta_im = uni((S).beta_im);               \
    (S).lambda = uni((S).lambda); (S).rel_tol = uni((S).rel_tol);                 \
    (S).iteration = uni((S).iteration); (S).max_iter = uni((S).max_iter);         \
    (S).done = uni((S).done); (S).pending = uni((S).pending);                     \
    (S).cur = uni((S).cur); (S).fresh = uni((S).fresh);                           \
  } while (0)
#define RLS_FISTA_UNIFORM(S)                                                      \
  do {                                                                            \
    (S).norm_x0 = uni((S).norm_x0); (S).res_norm = uni((S).res_norm);             \
    (S).rel_res_norm = uni((S).rel_res_norm); (S).rho = uni((S).rho);             \
    (S).theta = uni((S).theta); (S).theta_old = uni((S).theta_old);               \
    (S).rel_tol = uni((S).rel_tol); (S).lambda = uni((S).lambda);                 \
    (S).iteration = uni((S).iteration); (S).max_iter = uni((S).max_iter);         \
    (S).done = uni((S).done); (S).restart = uni((S).restart);                     \
    (S).reg_kind = uni((S).reg_kind); (S).proj_kind = uni((S).proj_kind);         \
    (S).l21_slices = uni((S).l21_slices); (S).pending = uni((S).pending);         \
    (S).ycur = uni((S).ycur); (S).fresh = uni((S).fresh);                         \
  } while (0)
#endif

// everything the FISTA pipeline kernels need (normal.hip)
struct rls_fista_pipe {
  const void* A;
  int64_t lda, M, N;
  void *b0, *b1;        // x / xold (caller's), swapped by iteration parity
  void *x0, *res;       // A^H b ; state.res (caller's)
  void *y0, *y1;        // extrapolated point, ping-pong (plan scratch)
  void *res_raw, *slab; // AHA y before "- x0" ; per-workgroup partial rows
  fista_scalars *sc, *scn;
  int par_hint = -1;    // parity of the iteration count (= which y / x buffer is current) at this launch, or -1;
                        // a hint checked on the device, like rls_cgnr_pipe::cur_hint
  rls_mailbox_slot mb;  // as rls_cgnr_pipe::mb
};
int32_t rls_fista_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P);
int32_t rls_fista_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P);
int32_t rls_fista_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P, void* sync, int n_steps,
                                  unsigned spin_limit, const rls_srv_args& Sv = rls_srv_args());

// everything the CGNR pipeline kernels need (normal.hip)
struct rls_cgnr_pipe {
  const void* A;
  int64_t lda, M, N;
  void *x, *r0, *p0, *r1, *p1, *v, *slab;
  double* dots;  // [nrhs][ndots][4]
  double* ttw = nullptr;  // [nrhs][row blocks]: K_A's ||t_w||^2 (alpha in its CGLS form; null: K_R forms <p, v>)
  int ndots;     // ceil(N / 16)
  cgnr_scalars *sc, *scn;  // [nrhs]
  int nrhs = 1;            // right-hand sides sharing one pass over A
  int64_t vstride = 0;     // elements between consecutive right-hand sides in x, r, p, v
  // which (r, p) pair is current when this launch runs, if the host knows (every step call starts at pair 0 and
  // each launch with a pending update flips it); -1 = unknown: the kernel then loads both candidates.  A hint
  // only: the kernel checks it against the device scalars and re-loads if it is wrong.
  int cur_hint = -1;
  rls_mailbox_slot mb;  // armed: the finish kernel publishes the scalars to the host itself (rls_cgnr_step_status)
};
int32_t rls_cgnr_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P);
int32_t rls_cgnr_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P);
int32_t rls_cgnr_pipe_launch(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P, int which);
// resident CGNR (normal.hip): the whole step call in one launch with A held in registers across iterations
size_t rls_cgnr_resident_sync_bytes();
// layout of the resident kernels' sync block: [0, clear_bytes) is zeroed ahead of every launch (arrival counters and the
// {fail, completed} words of that launch, which start at flags_offset); the sticky count of lost launches follows
size_t rls_resident_sync_alloc_bytes(int32_t dtype, int64_t N);  // sync block + the two-level exchange's group partials
size_t rls_resident_sync_clear_bytes();
size_t rls_resident_sync_flags_offset();   // {fail, completed, failed}: three consecutive unsigned words
size_t rls_resident_sync_placement_offset();  // "some workgroup is not on the XCD its group assumes": nonzero = partial rows written through
bool rls_cgnr_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda);
int rls_cgnr_resident_nwg(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N);
// cg! entry folded into a resident launch (src/ADMM.jl:236-244): r = b - (AHA + rho I) x with the warm start x, p = r,
// the scalars of the solve; with beta_y non-null b = beta_y + rho_admm (z - u) is formed on the way (and stored, with a
// copy of x in xold, by workgroup 0).  enabled = 0: the kernel continues an initialised solve (rls_cgnr_step).
struct rls_cg_start {
  int enabled = 0;
  const void* b = nullptr;
  const void *beta_y = nullptr, *z = nullptr, *u = nullptr;
  void *beta = nullptr, *xold = nullptr;
  float rho_admm = 0.f, rho = 0.f, reltol = 0.f;
  int maxiter = 0;
  const int* skip = nullptr;
  int* poison = nullptr;  // ADMM plans: set to 2 by a launch that gives up (resident_give_up), so that the kernels queued
                          // behind this cg! skip; the host re-runs the lost outer iterations from rls_admm_get_status
  // ---- server mode (rls_cgnr_step_status, the reference's solve! loop with callbacks: one iterate per call) ----------------------
  // srv_ctl != nullptr: after its step call the kernel does not end; workgroup 0 publishes the status (srv_mb), then LISTENS on a
  // control block in pinned host memory for the next command -- {sequence number, n_steps, mailbox sequence} -- and the grid runs
  // it from the registers it already holds: no launch, no load of A per call.  It leaves when nothing arrives for srv_idle_us
  // (or on an EXIT command: every other entry point of the library sends one before it touches the stream, rls_enter).
  // Layout (32-bit words): [0] command sequence  [1] n_steps  [2] mailbox sequence   (host writes)
  //                        [16] leaving  [17] exited: 1 = left idle / on EXIT, 2 = gave up inside a command   (device writes)
  unsigned* srv_ctl = nullptr;
  unsigned srv_seq0 = 0;      // the command sequence number this launch starts with (its n_steps is the launch argument)
  unsigned srv_idle_us = 0;
  rls_mailbox_slot srv_mb;
};

int32_t rls_cgnr_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_cgnr_pipe& P, double* dout, void* sync,
                                 int n_steps, unsigned spin_limit, const rls_cg_start& start = rls_cg_start());

// ---- OptISTA / POGM as resident launches (normal.hip, pgm_resident_kernel; host side pgm.hip) ----------------------------
// the 4-word device record of the deferred OptISTA / POGM sequences (rls_*_update_async): iteration count, `done`, ||res||
struct pgm_state {
  int iteration, done;
  float res_norm, pad;
};
// POGM with restart = :gradient (src/POGM.jl:183-232): theta, sigma, gamma travel in the device record behind pgm_state's words
struct pogm_auto_state {
  int iteration, done;
  float res_norm, pad;
  float theta, theta_old, sigma, gamma;
};
constexpr int RLS_PGM_MAX_IT = 48;  // iterations per resident launch: their coefficients travel as a kernel argument
struct rls_pgm_coefs {
  // per iteration, index-only scalars computed by the host in Float32 exactly as the reference does:
  //   OptISTA (src/OptISTA.jl:170-204): {rho gamma, rho gamma lambda, -1/gamma, 1/gamma, -beta, 1 + alpha + beta, -alpha, 0}
  //   POGM    (src/POGM.jl:183-210):    {rho, gamma lambda, c_y, c_x1, c_xo, c_z, 0, 0}
  // POGM with restart = :gradient (kind 2): the coefficients depend on the data (the restart decision), the kernel forms them
  // from the record (pogm_auto_state) as pogm_auto_kernel does; c[0] = {rho, lambda, sigma_fac, iterations of the solve}
  float c[RLS_PGM_MAX_IT][8];
};
struct rls_pgm_desc {
  const void* A;
  int64_t lda, M, N;
  int kind;                 // 0 = OptISTA, 1 = POGM (restart = :none), 2 = POGM (restart = :gradient)
  void *v0, *v1, *v2;       // loop-carried state: OptISTA x, y, z ; POGM x (operator input), y, z
  void* v3 = nullptr;       // kind 2: w (loop-carried as well)
  void *o0, *res;           // written every iteration, never read: OptISTA zold ; POGM xold ; and state.res
  const void* x0;
  void *slab, *raw;         // partial rows [nwg][N]; N-vector scratch of the flat exchange
  pgm_state* st;
  float norm_x0, rel_tol;
  int reg_kind, proj_kind;
  int first_it;             // st->iteration this launch expects (it does nothing otherwise)
};
bool rls_pgm_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda);
int32_t rls_pgm_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_pgm_desc& D, const rls_pgm_coefs& C, void* sync,
                                int n_steps, unsigned spin_limit);

// Gram-mode CGNR pipeline (normal.hip): one launch per iteration, every buffer in two parities
struct rls_gram_pipe {
  const void* G;
  int64_t ldg, N;
  void* x;
  void *r[2], *p[2], *v[2];   // index 0 = the caller's vectors, 1 = plan scratch
  double* dots;               // [2][nwg][4]
  cgnr_scalars* sc[2];
};
bool rls_gram_pipe_ok(int32_t dtype, int64_t N, const void* G, int64_t ldg);
int rls_gram_pipe_nwg(int32_t dtype, int64_t N);
int32_t rls_gram_pipe_iteration(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, int parity);
int32_t rls_gram_pipe_finish(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, int parity);
// resident Gram-mode CGNR / cg!: the whole step call as one launch, AHA held in registers (normal.hip)
bool rls_gram_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, const void* G, int64_t ldg);
bool rls_gram_resident_server_ok(int32_t dtype, int64_t N);
int32_t rls_gram_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_gram_pipe& P, void* sync, int n_steps,
                                 unsigned spin_limit, const rls_cg_start& start = rls_cg_start());

// Gram-mode FISTA pipeline (normal.hip)
struct rls_fista_gram {
  const void* G;
  int64_t ldg, N;
  void *b0, *b1, *x0, *res, *y0, *y1;
  void* rr[2];              // AHA y before "- x0", two parities
  fista_scalars* sc[2];
  int par_hint = -1;        // as rls_fista_pipe::par_hint
};
int32_t rls_fista_gram_iteration(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, int parity);
int32_t rls_fista_gram_finish(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, int parity);
int32_t rls_fista_gram_resident_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_gram& P, void* sync, int n_steps,
                                       unsigned spin_limit, const rls_srv_args& Sv = rls_srv_args());

// One right-hand side's slot in an MFMA operand panel (skinny.hip).  Full layout: panel[g][n][16] elements, column b in
// group b >> 4 at slot b & 15.  Half layout (complex, <= 8 columns): panel[n][16] floats = (re of 8 columns | im of 8).
#ifdef __HIPCC__
template <typename E>
struct panel_col {
  E* full;
  float* half;
  __device__ inline void put(int64_t i, E v) const {
    if (half) {
      half[16 * i] = elem<E>::re(v);
      half[16 * i + 8] = elem<E>::im(v);
    } else {
      full[16 * i] = v;
    }
  }
};
template <typename E>
__device__ static inline panel_col<E> panel_column(E* panel, int64_t n, int b, int half) {
  panel_col<E> c;
  c.full = half ? nullptr : panel + (int64_t)(b >> 4) * n * 16 + (b & 15);
  c.half = half ? reinterpret_cast<float*>(panel) + (b & 7) : nullptr;
  return c;
}
#endif

// everything the matrix-core batched kernels need (skinny.hip)
struct rls_skinny {
  const void* A;
  int64_t lda, M, N;
  const void* G = nullptr;    // explicit AHA (N x N, leading dimension ldg): V = G P is ONE product, same row splits
  int64_t ldg = 0;
  int nrhs, ngroups, splits;  // ngroups = ceil(nrhs / 16); splits = row splits of the A^H T product
  int half = 0;               // 1: complex, nrhs <= 8: ONE group whose 16 operand columns are (8 re | 8 im) floats
  void *X, *R, *P, *V;        // N x nrhs, columns ldv elements apart (caller's)
  int64_t ldv;
  float *Ppack, *Tpack;       // MFMA-operand layouts of P (N x 16 ngroups) and T (M x 16 ngroups)
  void* Vpart;                // [splits][16 ngroups (8 when half)][ldvp] partial A^H T
  int64_t ldvp;               // elements between right-hand sides in Vpart (N for the solver plans)
  cgnr_scalars* sc;           // [nrhs]
};
bool rls_skinny_ok(int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda);
// complex with at most 8 right-hand sides: the (re | im) operand packing (two MFMAs per complex block instead of four)
int rls_skinny_half(const rls_ctx* ctx, int32_t dtype, int nrhs);
static inline int rls_skinny_groups(int nrhs, int half) { return half ? 1 : (nrhs + 15) / 16; }
static inline int rls_skinny_pad(int nrhs, int half) { return half ? 8 : ((nrhs + 15) / 16) * 16; }
void rls_skinny_sizes(const rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, int nrhs, size_t* p_bytes, size_t* t_bytes,
                      size_t* v_bytes, int* splits);
int32_t rls_skinny_init(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, const void* B, int64_t ldb, float lambda,
                        float rel_tol, int max_iter);
int32_t rls_skinny_launch(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, int which);
int32_t rls_skinny_atb(rls_ctx* ctx, int32_t dtype, const rls_skinny& K, const void* B, int64_t ldb);
bool rls_gram_tiles_ok(int64_t M, int64_t N);
int32_t rls_gram_tiles(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G,
                       int64_t ldg);
int32_t rls_skinny_gram(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda, void* G,
                        int64_t ldg, void* panels);

// Small systems (M N s <= ~128 KiB): a whole rls_cgnr_step call as ONE single-workgroup launch, A in one CU's registers (small.hip)
struct rls_small {
  const void* A;
  int64_t lda, M, N;
  void *x, *r, *p, *v;
  cgnr_scalars* sc;
  rls_mailbox_slot mb;  // as rls_cgnr_pipe::mb
  // init! in the same launch (src/CGNR.jl:107-130): b != nullptr -> r = A^H b from the registers, x = v = 0, p = r, the scalars
  const void* b = nullptr;
  float lambda = 0.f, rel_tol = 0.f;
  int max_iter = 0;
  rls_srv_args srv;  // server mode (groups of one): the workgroup stays and listens for the next step call (rls_cg_start::srv_ctl)
};
// K independent small systems, one workgroup each, in ONE launch (the distinct-A flavour of a multi-solve,
// docs/src/literate/howto/multi_threading.jl:8-17): the descriptors travel as a kernel argument
constexpr int RLS_SMALL_GROUP_MAX = 24;
struct rls_small_group {
  int count = 0;
  rls_small d[RLS_SMALL_GROUP_MAX];
};
// the group travels BY VALUE as a kernel argument: the kernarg segment is 4 KiB, and the other arguments need a few words of it
static_assert(sizeof(rls_small_group) + 64 <= 4096, "rls_small_group no longer fits the 4 KiB kernarg segment: shrink rls_small or RLS_SMALL_GROUP_MAX");
int32_t rls_small_group_launch(rls_ctx* ctx, int32_t dtype, const rls_small_group& G, int n_steps);
bool rls_small_ok(int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda);
int32_t rls_small_launch(rls_ctx* ctx, int32_t dtype, const rls_small& D, int n_steps);
struct rls_fista_pipe;
int32_t rls_fista_small_launch(rls_ctx* ctx, int32_t dtype, const rls_fista_pipe& P, int n_steps,
                               const rls_srv_args& Sv = rls_srv_args());

// Batched CGNR on an explicit Gram matrix as ONE resident launch per step call (gramk.hip): up to 8 ComplexF32 right-hand
// sides, AHA (N <= 2048) held in the register files, the operand panel replicated in every workgroup's LDS
struct rls_gramk {
  const void* G;
  int64_t ldg, N;
  int nrhs;
  void *X, *R, *P, *V;  // N x nrhs, columns ldv elements apart (caller's)
  int64_t ldv;
  cgnr_scalars* sc;     // [nrhs]
  float* Vx;            // exchanged rows of V = AHA P, two parities (rls_gramk_sizes)
  void* Xx;             // x gathered at the end of the launch
  double* dots;         // per-workgroup partial <p, v>, ||p||^2, two parities
  float* Ppack;         // the streaming kernels' operand panel, kept in step (nullable)
};
void rls_gramk_sizes(int64_t N, size_t* vx_bytes, size_t* xx_bytes, size_t* dots_bytes);
bool rls_gramk_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, int nrhs, const void* G, int64_t ldg);
int32_t rls_gramk_resident_launch(rls_ctx* ctx, const rls_gramk& D, void* sync, int n_steps, unsigned spin_limit);
// the same for batched FISTA (no gradient restart; elementwise regularisers): rows of the next extrapolated point are what
// travels, the update itself is distributed (gramk.hip, fista_gramk_resident_kernel)
struct rls_fgramk {
  const void* G;
  int64_t ldg, N;
  int nrhs;
  void *b0, *b1, *x0, *res, *y;  // N x nrhs, columns ldv elements apart (x / xold by iteration parity, A^H b, residual, plan's y)
  int64_t ldv;
  fista_scalars* sc;    // [nrhs]
  float* Yx;            // exchanged rows of y, two parities (rls_fgramk_sizes)
  void* Xx;             // x, xold and res gathered at the end of the launch
  double* dots;         // per-workgroup partial ||res||^2, two parities
  float* Ypack;         // the streaming kernels' operand panel, kept in step (nullable)
};
void rls_fgramk_sizes(int64_t N, size_t* yx_bytes, size_t* xx_bytes, size_t* dots_bytes);
bool rls_fgramk_resident_ok(rls_ctx* ctx, int32_t dtype, int64_t N, int nrhs, const void* G, int64_t ldg);
int32_t rls_fgramk_resident_launch(rls_ctx* ctx, const rls_fgramk& D, void* sync, int n_steps, unsigned spin_limit);

// ---------------------------------------------------------------------------------------------
// comm.hip internals used by the row-sharded solver loops (solvers.hip)
// ---------------------------------------------------------------------------------------------
#include "host_pool.hpp"  // rls_comm_phase, the worker pool, the pinned cache, the resident-chain bookkeeping (device-free)
int32_t rls_comm_run(rls_comm* c, const std::vector<rls_comm_phase>& phases, int reps);
int32_t rls_comm_publish(rls_comm* c, int rank, void* buf, int64_t n, int32_t dtype, int round);
int32_t rls_comm_collect(rls_comm* c, int rank, void* buf, int64_t n, int32_t dtype, int round);
int32_t rls_comm_next_rounds(rls_comm* c, int count, int64_t n, int32_t dtype, int* first);

// ---------------------------------------------------------------------------------------------
// host-side launch entry points implemented in the .hip files (all enqueue on ctx->stream)
// ---------------------------------------------------------------------------------------------
// gemv.hip.  `skip` (nullable) is a device int: when non-zero at kernel entry the kernel is a no-op
int32_t rls_launch_gemv(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, float ar, float ai,
                        const void* A, int64_t lda, const void* x, float br, float bi, void* y, const int* skip);
// normal.hip: v = A^H A p in ONE pass over A (slab of A held in registers between the two products).
// Returns the slab workspace size in bytes (0 = shape not supported by the fused kernel).
// tv.hip: the FGP loop as ONE single-workgroup launch, out = prox_TV(xin [+ add]) (`add`, `skip` nullable)
bool rls_tv_single_ok(const rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv, const int32_t* dims);
int32_t rls_tv_single_launch(rls_ctx* ctx, int32_t dtype, int32_t ndims, const int64_t* shape, int32_t ntv,
                             const int32_t* dims, const void* xin, const void* add, void* out, float lam, int iters,
                             const int* skip, int count = 1, int64_t ldv = 0, int skip_stride = 0);
size_t rls_normal_fused_workspace(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda);
int32_t rls_launch_normal_fused(rls_ctx* ctx, int32_t dtype, int64_t M, int64_t N, const void* A, int64_t lda,
                                const void* p, void* v, void* slab, const int* skip);
