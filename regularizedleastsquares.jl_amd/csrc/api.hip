// Context, memory and timing entry points of the C ABI (include/rls_mi355x.h).
#include "rls_common.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <set>
#include <vector>

// ---- memory (rls_common.hpp) -------------------------------------------------------------------------------------------
namespace {
std::mutex g_mem_mutex;
std::map<const rls_ctx*, uint64_t> g_live_ctx;  // live contexts and their generation ids
uint64_t g_next_ctx_id = 1;
thread_local rls_ctx* tl_alloc_ctx = nullptr;

// One PRIVATE stream-ordered pool per device (release threshold raised so that freed blocks stay cached).  The device's default
// pool is shared with every other hipMallocAsync user of the process (PyTorch, AMDGPU.jl): its settings are not ours to change.
std::map<int, hipMemPool_t> g_pools;
hipMemPool_t device_pool(int device) {
  static std::mutex m;
  std::lock_guard<std::mutex> lk(m);
  auto it = g_pools.find(device);
  if (it != g_pools.end()) return it->second;
  hipMemPool_t pool = nullptr;
  const char* env = getenv("RLS_ALLOC");
  if (!(env && !strcmp(env, "sync"))) {
    int supported = 0;
    if (hipDeviceGetAttribute(&supported, hipDeviceAttributeMemoryPoolsSupported, device) == hipSuccess && supported) {
      hipMemPoolProps props;
      memset(&props, 0, sizeof(props));
      props.allocType = hipMemAllocationTypePinned;
      props.handleTypes = hipMemHandleTypeNone;
      props.location.type = hipMemLocationTypeDevice;
      props.location.id = device;
      if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
        uint64_t keep = ~0ull;  // never hand cached blocks back to the driver at synchronisation points
        if (hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep) != hipSuccess) {
          (void)hipMemPoolDestroy(pool);
          pool = nullptr;
        }
      } else {
        pool = nullptr;
      }
    }
    (void)hipGetLastError();
  }
  g_pools[device] = pool;
  return pool;
}
bool device_pools_ok(int device) { return device_pool(device) != nullptr; }
}  // namespace

// by address only: for entry points that receive nothing but the handle (rls_free from a garbage-collected host's finalizer)
static bool ctx_alive_by_address(const rls_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  return ctx && g_live_ctx.find(ctx) != g_live_ctx.end();
}

bool rls_ctx_alive(const rls_ctx* ctx, uint64_t id) {
  std::lock_guard<std::mutex> lk(g_mem_mutex);
  if (!ctx) return false;
  const auto it = g_live_ctx.find(ctx);
  return it != g_live_ctx.end() && it->second == id;
}

hipError_t rls_dev_alloc(rls_ctx* ctx, void** p, size_t bytes) {
  if (ctx && ctx->pools) return hipMallocFromPoolAsync(p, bytes ? bytes : 1, device_pool(ctx->device), ctx->stream);
  return hipMalloc(p, bytes ? bytes : 1);
}
hipError_t rls_dev_free(rls_ctx* ctx, void* p) {
  if (!p) return hipSuccess;
  if (ctx && ctx->pools) return hipFreeAsync(p, ctx->stream);
  return hipFree(p);
}

// small pinned host blocks: the process-wide cache of host_pool.hpp (keeps blocks <= 4 KiB, returns larger ones to the driver).
// mapped + coherent: the status kernels store into these blocks directly (rls_fetch_wait)
static pinned_cache g_pinned;
hipError_t rls_pinned_alloc(void** p, size_t bytes) {
  hipError_t err = hipSuccess;
  *p = g_pinned.get(bytes, [&err](size_t n) -> void* {
    void* raw = nullptr;
    err = hipHostMalloc(&raw, n, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    return err == hipSuccess ? raw : nullptr;
  });
  return *p ? hipSuccess : (err != hipSuccess ? err : hipErrorOutOfMemory);
}
void rls_pinned_free(void* p) {
  g_pinned.put(p, [](void* raw) { (void)hipHostFree(raw); });
}

// ---- status mailbox --------------------------------------------------------------------------------------------------------
struct fetch_args {
  const unsigned* src[RLS_FETCH_MAX];
  unsigned* dst[RLS_FETCH_MAX];
  unsigned n[RLS_FETCH_MAX];
  int count;
};
// one wave: every queued block, dword by dword, straight into pinned host memory (system-scope stores), then -- released
// behind them -- the sequence word the host is spinning on
__global__ __launch_bounds__(64) void mailbox_publish_kernel(fetch_args A, unsigned* seq_h, unsigned seq) {
  for (int k = 0; k < A.count; ++k)
    for (unsigned i = threadIdx.x; i < A.n[k]; i += 64)
      __hip_atomic_store(A.dst[k] + i, A.src[k][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  rls_system_stores_done();  // the wave's stores above are acknowledged ...
  if (threadIdx.x == 0) __hip_atomic_store(seq_h, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // ... before this one goes out
}

int32_t rls_fetch_add(rls_ctx* ctx, const void* src_d, void* dst_pinned, size_t bytes) {
  if (!ctx->tune.status_mailbox || (bytes & 3) || ctx->nfq >= RLS_FETCH_MAX) {
    RLS_HIP(ctx, hipMemcpyAsync(dst_pinned, src_d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
  }
  ctx->fq[ctx->nfq++] = {src_d, dst_pinned, (unsigned)(bytes / 4)};
  return 0;
}

int32_t rls_fetch_wait(rls_ctx* ctx) {
  if (ctx->nfq == 0) {
    RLS_HIP(ctx, rls_stream_wait(ctx->stream));
    return 0;
  }
  fetch_args A;
  A.count = ctx->nfq;
  for (int k = 0; k < RLS_FETCH_MAX; ++k) {
    A.src[k] = k < ctx->nfq ? reinterpret_cast<const unsigned*>(ctx->fq[k].src) : nullptr;
    A.dst[k] = k < ctx->nfq ? reinterpret_cast<unsigned*>(ctx->fq[k].dst) : nullptr;
    A.n[k] = k < ctx->nfq ? ctx->fq[k].dwords : 0u;
  }
  ctx->nfq = 0;
  const unsigned seq = ++ctx->mb_seq;
  hipLaunchKernelGGL(mailbox_publish_kernel, dim3(1), dim3(64), 0, ctx->stream, A, ctx->mb_h, seq);
  RLS_HIP(ctx, hipGetLastError());
  return rls_mailbox_wait(ctx, seq);
}

rls_mailbox_slot rls_mailbox_arm(rls_ctx* ctx, void* dst_pinned) {
  rls_mailbox_slot mb;
  if (ctx->tune.status_mailbox < 2) return mb;
  mb.dst = dst_pinned;
  mb.seq_h = ctx->mb_h;
  mb.seq = ++ctx->mb_seq;
  return mb;
}

int32_t rls_mailbox_wait(rls_ctx* ctx, unsigned seq) {
  volatile unsigned* p = ctx->mb_h;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned n = 0;; ++n) {
    if (*p == seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return 0;
    }
    rls_cpu_relax();
    if ((n & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(250)) break;
  }
  // a long queue in front of the kernel, or a fault: block on the stream (which reports the error) and look again
  RLS_HIP(ctx, hipStreamSynchronize(ctx->stream));
  std::atomic_thread_fence(std::memory_order_acquire);
  return *p == seq ? 0 : rls_fail(ctx, RLS_E_STATE, "status mailbox: the publishing kernel did not run");
}

rls_alloc_scope::rls_alloc_scope(rls_ctx* ctx) : prev(tl_alloc_ctx) { tl_alloc_ctx = ctx; }
rls_alloc_scope::~rls_alloc_scope() { tl_alloc_ctx = prev; }
hipError_t rls_scoped_malloc(void** p, size_t bytes) { return rls_dev_alloc(tl_alloc_ctx, p, bytes); }
hipError_t rls_scoped_free(void* p) { return rls_dev_free(tl_alloc_ctx, p); }

static int32_t ctx_setup(rls_ctx* ctx) {
  RLS_HIP(ctx, hipEventCreate(&ctx->ev0));
  RLS_HIP(ctx, hipEventCreate(&ctx->ev1));
  RLS_HIP(ctx, hipMalloc((void**)&ctx->red_d, sizeof(double) * RLS_RED_SLOTS));
  RLS_HIP(ctx, hipMalloc((void**)&ctx->res_d, sizeof(float) * RLS_RES_FLOATS));
  RLS_HIP(ctx, hipHostMalloc((void**)&ctx->res_h, sizeof(float) * RLS_RES_FLOATS, hipHostMallocDefault));
  RLS_HIP(ctx, hipHostMalloc((void**)&ctx->mb_h, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
  ctx->mb_h[0] = 0;
  return 0;
}

static int32_t ctx_create_impl(int32_t device, void* stream, bool borrow, rls_ctx** out) {
  if (!out) return RLS_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess) return (int32_t)e;
  if (device < 0 || device >= ndev) return RLS_E_INVALID;
  e = hipSetDevice(device);
  if (e != hipSuccess) return (int32_t)e;
  rls_ctx* ctx = new rls_ctx();
  ctx->device = device;
  ctx->pools = device_pools_ok(device);
  if (borrow) {
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
    // A kernel left listening keeps the state it wrote in its XCD's L2 until it leaves, and only entry points of THIS context
    // tell it to leave.  The owner of a borrowed stream reads the state vectors by its own means (another library, another
    // stream behind an event), so such contexts start with the mode off; rls_tune_set("resident_server", 1) opts in.
    ctx->tune.resident_server = 0;
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return (int32_t)e;
    }
    ctx->own_stream = true;
  }
  int32_t st = ctx_setup(ctx);
  if (st != 0) {
    rls_ctx_destroy(ctx);
    return st;
  }
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    ctx->id = g_next_ctx_id++;
    g_live_ctx[ctx] = ctx->id;
  }
  *out = ctx;
  return 0;
}

extern "C" {

int32_t rls_abi_version(void) { return RLS_ABI_VERSION; }

int32_t rls_device_count(int32_t* out) {
  if (!out) return RLS_E_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *out = (e == hipSuccess) ? n : 0;
  return (int32_t)e;
}

int32_t rls_ctx_create(int32_t device, rls_ctx** out) { return ctx_create_impl(device, nullptr, false, out); }
int32_t rls_ctx_create_on_stream(int32_t device, void* hip_stream, rls_ctx** out) {
  return ctx_create_impl(device, hip_stream, true, out);
}

int32_t rls_ctx_destroy(rls_ctx* ctx) {
  RLS_CHECK_CTX(ctx);
  {
    std::lock_guard<std::mutex> lk(g_mem_mutex);
    g_live_ctx.erase(ctx);
  }
  rls_enter(ctx);
  if (ctx->stream) rls_stream_wait(ctx->stream);
  rls_resident_forget(ctx->device, ctx->stream);  // (after the wait: nothing of this stream is in flight any more)
  if (ctx->ev0) hipEventDestroy(ctx->ev0);
  if (ctx->ev1) hipEventDestroy(ctx->ev1);
  if (ctx->red_d) hipFree(ctx->red_d);
  if (ctx->res_d) hipFree(ctx->res_d);
  if (ctx->res_h) hipHostFree(ctx->res_h);
  if (ctx->mb_h) hipHostFree(ctx->mb_h);
  if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return 0;
}

int32_t rls_ctx_sync(rls_ctx* ctx) {
  RLS_CHECK_CTX(ctx);
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  return 0;
}

void* rls_ctx_stream(rls_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
const char* rls_last_error_string(rls_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int32_t rls_tune_set(rls_ctx* ctx, const char* key, int32_t value) {
  RLS_CHECK_CTX(ctx);
  if (!key) return RLS_E_INVALID;
  if (ctx->server) rls_server_stop(ctx);  // (a kernel left listening was launched under the old settings)
  ++ctx->tune_epoch;                       // (... and so was every cached graph: run_steps captures afresh)
  if (!strcmp(key, "gemvn_g")) ctx->tune.gemvn_g = value;
  else if (!strcmp(key, "gemvn_waves")) ctx->tune.gemvn_waves = value;
  else if (!strcmp(key, "gemvt_cols")) ctx->tune.gemvt_cols = value;
  else if (!strcmp(key, "gemvt_reverse")) ctx->tune.gemvt_reverse = value;
  else if (!strcmp(key, "graph_chunk")) ctx->tune.graph_chunk = value;
  else if (!strcmp(key, "use_graph")) ctx->tune.use_graph = value;
  else if (!strcmp(key, "fuse_level")) ctx->tune.fuse_level = value;
  else if (!strcmp(key, "fused_normal")) ctx->tune.fused_normal = value;
  else if (!strcmp(key, "cgnr_pipeline")) ctx->tune.cgnr_pipeline = value;
  else if (!strcmp(key, "batched_mfma")) ctx->tune.batched_mfma = value;
  else if (!strcmp(key, "gram_pipeline")) ctx->tune.gram_pipeline = value;
  else if (!strcmp(key, "pipe_hint_mode")) ctx->tune.pipe_hint_mode = value;
  else if (!strcmp(key, "resident")) {
    ctx->tune.resident = value;
    ctx->resident_failures = 0;  // an explicit switch also forgets earlier timeouts (solvers.hip, resident_lost)
  }
  else if (!strcmp(key, "status_mailbox")) ctx->tune.status_mailbox = value;
  else if (!strcmp(key, "small")) ctx->tune.small = value;
  else if (!strcmp(key, "resident_server")) ctx->tune.resident_server = value;
  else if (!strcmp(key, "resident_ahead")) ctx->tune.resident_ahead = value ? 1 : 0;
  else if (!strcmp(key, "resident_l2_rows")) ctx->tune.resident_l2_rows = value;
  else if (!strcmp(key, "fista_defer")) ctx->tune.fista_defer = value;
  else if (!strcmp(key, "resident_server_idle_us")) {
    // the other workgroups of a listening grid wait at a barrier whose bound is resident_spin polls (about 0.1 s at the default):
    // an idle time beyond a fraction of that would make them give up while workgroup 0 still polls the host (a lost launch)
    ctx->tune.resident_server_idle_us = value < 1 ? 1 : (value > 10000 ? 10000 : value);
  }
  else if (!strcmp(key, "resident_spin")) ctx->tune.resident_spin = value;
  else if (!strcmp(key, "resident_preclear")) ctx->tune.resident_preclear = value;
  else if (!strcmp(key, "skinny_t_waves")) ctx->tune.skinny_t_waves = value;
  else if (!strcmp(key, "skinny_v_waves")) ctx->tune.skinny_v_waves = value;
  else if (!strcmp(key, "skinny_v_splits")) ctx->tune.skinny_v_splits = value;
  else if (!strcmp(key, "kaczmarz_nt")) ctx->tune.kaczmarz_nt = value;
  else if (!strcmp(key, "skinny_t_u")) ctx->tune.skinny_t_u = value;
  else if (!strcmp(key, "skinny_v_u")) ctx->tune.skinny_v_u = value;
  else if (!strcmp(key, "skinny_half")) ctx->tune.skinny_half = value;
  else if (!strcmp(key, "skinny_t_roll")) ctx->tune.skinny_t_roll = value;
  else if (!strcmp(key, "skinny_v_roll")) ctx->tune.skinny_v_roll = value;
  else if (!strcmp(key, "skinny_g_roll")) ctx->tune.skinny_g_roll = value;
  else if (!strcmp(key, "gram_lds_kib")) ctx->tune.gram_lds = value * 1024;
  else if (!strcmp(key, "slab_g")) ctx->tune.slab_g = value;  // must be set before the operator is created (its workspace is sized by it)
  else if (!strcmp(key, "slab_wv")) {  // (only 8-wave slabs are instantiated since round 3: the switch is kept for old scripts)
    if (value != 0 && value != 8) return rls_fail(ctx, RLS_E_INVALID, "tune_set: slab_wv: only 8-wave slabs exist");
  }
  else if (!strcmp(key, "tv_fused_max_n")) ctx->tune.tv_fused_max_n = value;
  else if (!strcmp(key, "tv_fused_2d")) ctx->tune.tv_fused_2d = value;
  else if (!strcmp(key, "slab_order")) ctx->tune.slab_order = value;
  else if (!strcmp(key, "red_threads")) ctx->tune.red_threads = value;
  else if (!strcmp(key, "slab_multi")) ctx->tune.slab_multi = value ? 1 : 0;
  else if (!strcmp(key, "resident_barrier")) ctx->tune.resident_barrier = value == 1 ? 1 : 2;
  else return rls_fail(ctx, RLS_E_INVALID, "tune_set: unknown key");
  return 0;
}

int32_t rls_malloc(rls_ctx* ctx, size_t bytes, void** out) {
  RLS_CHECK_CTX(ctx);
  if (!out) return rls_fail(ctx, RLS_E_INVALID, "malloc: null out");
  *out = nullptr;
  if (bytes == 0) return 0;
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, rls_dev_alloc(ctx, out, bytes));
  return 0;
}

int32_t rls_free(rls_ctx* ctx, void* p) {
  if (!p) return 0;
  // A garbage-collected host runs finalizers in no particular order: an array allocated on a communicator's context can be
  // freed after rls_comm_destroy has destroyed that context (julia/RLSMI355X: Comm and the arrays of its shards).  A handle that
  // is not a LIVE context is never dereferenced: the block goes back synchronously (hipFree resolves the device from the pointer).
  if (!ctx_alive_by_address(ctx)) {
    const hipError_t e = hipFree(p);
    return e == hipSuccess ? 0 : (int32_t)e;
  }
  if (ctx->pools && ctx->server) {
    // a kernel left listening (server mode): the pooled free is ordered behind it on the stream by itself and does not block the host.
    // It must NOT ask that kernel to leave: frees come from the host's garbage collector at any time (a finalizer, Python's cycle
    // collector), and two kernel lives that short in a row put the plan on the per-iteration pipeline for the rest of its solve.
    RLS_HIP(ctx, hipSetDevice(ctx->device));
    RLS_HIP(ctx, rls_dev_free(ctx, p));
    return 0;
  }
  RLS_HIP(ctx, rls_enter(ctx));
  if (!ctx->pools) RLS_HIP(ctx, rls_stream_wait(ctx->stream));  // (hipFree synchronises the whole device anyway)
  RLS_HIP(ctx, rls_dev_free(ctx, p));   // pooled: ordered behind everything enqueued on the context's stream
  return 0;
}

int32_t rls_memcpy_h2d(rls_ctx* ctx, void* dst, const void* src_h, size_t bytes) {
  RLS_CHECK_CTX(ctx);
  if (bytes == 0) return 0;
  if (!dst || !src_h) return rls_fail(ctx, RLS_E_INVALID, "memcpy_h2d: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  // pageable host memory: the copy is staged, so wait for it before the caller may reuse src_h
  RLS_HIP(ctx, hipMemcpyAsync(dst, src_h, bytes, hipMemcpyHostToDevice, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  return 0;
}

int32_t rls_memcpy_d2h(rls_ctx* ctx, void* dst_h, const void* src, size_t bytes) {
  RLS_CHECK_CTX(ctx);
  if (bytes == 0) return 0;
  if (!dst_h || !src) return rls_fail(ctx, RLS_E_INVALID, "memcpy_d2h: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, hipMemcpyAsync(dst_h, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RLS_HIP(ctx, rls_stream_wait(ctx->stream));
  return 0;
}

int32_t rls_memcpy_d2d(rls_ctx* ctx, void* dst, const void* src, size_t bytes) {
  RLS_CHECK_CTX(ctx);
  if (bytes == 0) return 0;
  if (!dst || !src) return rls_fail(ctx, RLS_E_INVALID, "memcpy_d2d: null pointer");
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

int32_t rls_timer_start(rls_ctx* ctx) {
  RLS_CHECK_CTX(ctx);
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  return 0;
}

int32_t rls_timer_stop_ms(rls_ctx* ctx, float* ms_out) {
  RLS_CHECK_CTX(ctx);
  if (!ms_out) return rls_fail(ctx, RLS_E_INVALID, "timer_stop: null out");
  RLS_HIP(ctx, rls_enter(ctx));
  RLS_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  RLS_HIP(ctx, rls_event_wait(ctx->ev1));
  RLS_HIP(ctx, hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return 0;
}

int32_t rls_gemv(rls_ctx* ctx, int32_t dtype, int32_t op, int64_t M, int64_t N, float alpha_re, float alpha_im,
                 const void* A, int64_t lda, const void* x, float beta_re, float beta_im, void* y) {
  return rls_launch_gemv(ctx, dtype, op, M, N, alpha_re, alpha_im, A, lda, x, beta_re, beta_im, y, nullptr);
}

}  // extern "C"
