// Communicator for the row-partitioned mode (BASELINE config 5): one exchange step per operator
// apply -- the all-reduce(sum) of the length-N partial products A_g^H t_g -- between the ranks of ONE host process
// (the Julia host drives every GPU of the node from one process, one task per GPU: src/MultiThreading.jl:60-78 is the
// fan-out site this serves; the Python host of this repository runs one process per GPU and uses torch.distributed).
//
// Two transports behind the same entry point:
//   * RLS_COMM_RCCL   -- ncclAllReduce on every rank's stream inside one group call (RCCL over xGMI).  The library is
//                        loaded with dlopen at communicator creation, so librls_mi355x.so itself does not link it.
//                        Needs distinct devices (RCCL rejects two ranks on one GPU).
//   * RLS_COMM_DIRECT -- the one-shot direct-write all-reduce for the 64 KiB vectors of this path (latency-bound at
//                        that size: SURVEY 5, last row): every rank stores its vector straight into slot r of every
//                        peer's receive buffer (peer access: xGMI stores, or plain stores when ranks share a device),
//                        records an event, waits for the events of all ranks on its own stream and sums the slots
//                        in RANK ORDER.  Every rank adds the same numbers in the same order, so the replicated state
//                        of the solvers stays bit-identical across ranks by construction (a ring all-reduce gives
//                        every rank the same bits too, but not the bits of the unsharded sum order).  Ordering is
//                        by stream events only -- no in-kernel polling across devices -- and receive buffers
//                        alternate between two parities so that round k + 1 never overwrites what round k still reads.
//                        Ranks may share a device: that is how the collective schedule of config 5 runs on a
//                        one-GPU box (tests/abi_smoke.c, tests/test_gpu_parity.py).
#include "rls_common.hpp"

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = RLS_POOL_MAX_RANKS;

struct peer_ptrs {
  float* p[MAX_RANKS];
};

// src -> slot `rank` of every rank's receive buffer (blockIdx.y = destination rank)
__global__ __launch_bounds__(256) void comm_push_kernel(const float* __restrict__ src, peer_ptrs dst, int64_t nf) {
  float* out = dst.p[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (int64_t)gridDim.x * blockDim.x) out[i] = src[i];
}

// buf = slot 0 + slot 1 + ... + slot n-1, in that order
__global__ __launch_bounds__(256) void comm_sum_kernel(float* __restrict__ buf, const float* __restrict__ slots, int n,
                                                       int64_t stride_f, int64_t nf) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf; i += (int64_t)gridDim.x * blockDim.x) {
    float s = slots[i];
    for (int r = 1; r < n; ++r) s += slots[(int64_t)r * stride_f + i];
    buf[i] = s;
  }
}

typedef void* nccl_comm_t;
struct rccl_api {
  void* handle = nullptr;
  int (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

}  // namespace

struct rls_comm {
  int n = 0;
  int transport = 0;
  std::vector<rls_ctx*> ctx;
  bool own_ctx = false;
  // direct transport
  size_t cap_f = 0;                  // floats per slot
  std::vector<float*> recv[2];       // [parity][rank]: n slots of cap_f floats on that rank's device
  std::vector<hipEvent_t> pushed[2]; // [parity][rank]
  int round = 0;
  // host fan-out
  int use_threads = 1;
  comm_pool* pool = nullptr;
  // RCCL transport
  rccl_api rccl;
  std::vector<nccl_comm_t> comms;
  bool group_open = false;  // single-thread path: rank 0 opened an RCCL group that the last rank has not closed yet
  // peer-access probe (rls_comm_peer_access): peer[r * n + t] = 1 rank r's device can store into rank t's (or they share a
  // device), 0 it cannot, -1 the query itself failed; requested = the transport the caller asked for, transport = the one in use
  std::vector<int32_t> peer;
  int requested = 0;
};


static int32_t comm_fail(rls_comm* c, int32_t code, const char* what) {
  return rls_fail(c && !c->ctx.empty() ? c->ctx[0] : nullptr, code, what);
}

static int32_t direct_reserve(rls_comm* c, size_t nf) {
  if (nf <= c->cap_f) return 0;
  rls_ctx* c0 = c->ctx[0];
  for (int r = 0; r < c->n; ++r) {  // growing the receive buffers is a setup step: drain every rank first
    RLS_HIP(c0, hipSetDevice(c->ctx[r]->device));
    RLS_HIP(c0, rls_stream_wait(c->ctx[r]->stream));
  }
  const size_t cap = (nf + 1023) / 1024 * 1024;
  for (int q = 0; q < 2; ++q)
    for (int r = 0; r < c->n; ++r) {
      RLS_HIP(c0, hipSetDevice(c->ctx[r]->device));
      if (c->recv[q][r]) RLS_HIP(c0, hipFree(c->recv[q][r]));
      c->recv[q][r] = nullptr;
      RLS_HIP(c0, hipMalloc((void**)&c->recv[q][r], sizeof(float) * cap * (size_t)c->n));
    }
  c->cap_f = cap;
  return 0;
}

// rank r's half-steps of the direct all-reduce for round `round` (parity round & 1):
//   publish: r's vector into slot r of EVERY rank's receive buffer, then r's "pushed" event;
//   collect: wait for every other rank's "pushed" event of this round on r's stream, then sum the slots in rank order.
// Every rank's publish (the event record) must have been CALLED before any rank's collect is called: a host barrier
// between the two when the ranks run on worker threads, the loop structure when one thread drives them all.
static int32_t direct_publish(rls_comm* c, int r, const void* buf, size_t nf, int round) {
  rls_ctx* cr = c->ctx[r];
  const int q = round & 1;
  const unsigned gx = (unsigned)((nf + 255) / 256 < 64 ? (nf + 255) / 256 : 64);
  RLS_HIP(cr, hipSetDevice(cr->device));
  peer_ptrs d;
  for (int t = 0; t < MAX_RANKS; ++t) d.p[t] = t < c->n ? c->recv[q][t] + (size_t)r * c->cap_f : nullptr;
  hipLaunchKernelGGL(comm_push_kernel, dim3(gx, (unsigned)c->n), dim3(256), 0, cr->stream, (const float*)buf, d, (int64_t)nf);
  RLS_HIP(cr, hipEventRecord(c->pushed[q][r], cr->stream));
  return 0;
}
static int32_t direct_collect(rls_comm* c, int r, void* buf, size_t nf, int round) {
  rls_ctx* cr = c->ctx[r];
  const int q = round & 1;
  const unsigned gx = (unsigned)((nf + 255) / 256 < 64 ? (nf + 255) / 256 : 64);
  RLS_HIP(cr, hipSetDevice(cr->device));
  for (int t = 0; t < c->n; ++t)
    if (t != r) RLS_HIP(cr, hipStreamWaitEvent(cr->stream, c->pushed[q][t], 0));
  hipLaunchKernelGGL(comm_sum_kernel, dim3(gx), dim3(256), 0, cr->stream, (float*)buf, (const float*)c->recv[q][r], c->n,
                     (int64_t)c->cap_f, (int64_t)nf);
  RLS_HIP(cr, hipGetLastError());
  return 0;
}

static int32_t direct_allreduce(rls_comm* c, void* const* bufs, size_t nf) {
  RLS_TRY(direct_reserve(c, nf));
  const int round = c->round++;
  for (int r = 0; r < c->n; ++r) RLS_TRY(direct_publish(c, r, bufs[r], nf, round));
  for (int r = 0; r < c->n; ++r) RLS_TRY(direct_collect(c, r, bufs[r], nf, round));
  return 0;
}

static int32_t rccl_load(rls_comm* c) {
  rccl_api& R = c->rccl;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (R.handle) break;
  }
  if (!R.handle) return comm_fail(c, RLS_E_UNSUPPORTED, "rls_comm_create: librccl.so not found (RCCL transport)");
  R.CommInitAll = (int (*)(nccl_comm_t*, int, const int*))dlsym(R.handle, "ncclCommInitAll");
  R.CommDestroy = (int (*)(nccl_comm_t))dlsym(R.handle, "ncclCommDestroy");
  R.GroupStart = (int (*)())dlsym(R.handle, "ncclGroupStart");
  R.GroupEnd = (int (*)())dlsym(R.handle, "ncclGroupEnd");
  R.AllReduce = (int (*)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t))dlsym(R.handle, "ncclAllReduce");
  R.GetErrorString = (const char* (*)(int))dlsym(R.handle, "ncclGetErrorString");
  if (!R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.AllReduce)
    return comm_fail(c, RLS_E_UNSUPPORTED, "rls_comm_create: librccl.so lacks an expected symbol");
  return 0;
}

static int32_t rccl_check(rls_comm* c, int rc, const char* what) {
  if (rc == 0) return 0;
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", what, c->rccl.GetErrorString ? c->rccl.GetErrorString(rc) : "RCCL error");
  return comm_fail(c, 1000 + rc, msg);
}

// ---- host fan-out: the worker pool and its barrier live in host_pool.hpp (device-free: also built under the sanitizers) ----
// reps x (the phases in order) for every rank.  Threads: each rank's worker runs the whole sequence, meeting the others
// at the host barrier behind every phase that asks for one.  One thread (single rank, or threads switched off): phase by
// phase over all ranks, which orders everything a barrier would.
int32_t rls_comm_run(rls_comm* c, const std::vector<rls_comm_phase>& phases, int reps) {
  if (!c) return RLS_E_INVALID;
  if (reps <= 0 || phases.empty()) return 0;
  if (c->n == 1 || !c->use_threads) {
    for (int k = 0; k < reps; ++k)
      for (const rls_comm_phase& ph : phases)
        for (int r = 0; r < c->n; ++r) {
          const int32_t st = ph.run(r, k);
          if (st != 0) {
            // a phase body failed between rank 0's ncclGroupStart and the last rank's ncclGroupEnd: close the group, or every
            // later RCCL call of this thread would be queued into it
            if (c->group_open) {
              (void)c->rccl.GroupEnd();
              c->group_open = false;
            }
            return st;
          }
        }
    return 0;
  }
  rls_comm* cc = c;
  return pool_run(c->pool, c->n, phases, reps, [cc](int r) { (void)hipSetDevice(cc->ctx[r]->device); });
}

// the collective as two per-rank half-steps (see direct_publish / direct_collect); `round` = rls_comm_next_rounds() + k.
// RCCL: each rank's ncclAllReduce on its own communicator from its own thread (no group needed), collect is a no-op;
// when ONE thread drives all ranks the calls of a round sit inside one group (rank 0 opens it, the last rank closes it).
int32_t rls_comm_publish(rls_comm* c, int r, void* buf, int64_t n, int32_t dtype, int round) {
  const size_t nf = (size_t)n * (dtype == RLS_C32 ? 2 : 1);
  if (c->n == 1 || n == 0) return 0;
  if (c->transport == RLS_COMM_DIRECT) return direct_publish(c, r, buf, nf, round);
  const bool grouped = !c->use_threads;
  if (grouped && r == 0) {
    RLS_TRY(rccl_check(c, c->rccl.GroupStart(), "ncclGroupStart"));
    c->group_open = true;
  }
  const int32_t st = rccl_check(c, c->rccl.AllReduce(buf, buf, nf, /* ncclFloat32 */ 7, /* ncclSum */ 0, c->comms[r], c->ctx[r]->stream),
                                "ncclAllReduce");
  if (grouped && r == c->n - 1) {
    const int32_t st2 = rccl_check(c, c->rccl.GroupEnd(), "ncclGroupEnd");
    c->group_open = false;
    return st != 0 ? st : st2;
  }
  return st;
}
int32_t rls_comm_collect(rls_comm* c, int r, void* buf, int64_t n, int32_t dtype, int round) {
  const size_t nf = (size_t)n * (dtype == RLS_C32 ? 2 : 1);
  if (c->n == 1 || n == 0 || c->transport != RLS_COMM_DIRECT) return 0;
  return direct_collect(c, r, buf, nf, round);
}
// reserve `count` consecutive round numbers (and the receive buffers for vectors of n elements) ahead of a run
int32_t rls_comm_next_rounds(rls_comm* c, int count, int64_t n, int32_t dtype, int* first) {
  if (c->transport == RLS_COMM_DIRECT && c->n > 1) RLS_TRY(direct_reserve(c, (size_t)n * (dtype == RLS_C32 ? 2 : 1)));
  *first = c->round;
  c->round += count;
  return 0;
}

extern "C" {

// measurement (tools/host_overhead_rowsharded.py): seconds each rank's worker has spent enqueueing since the last call
// (out[nranks]); what the host side of the row-sharded loops would cost on a node where every rank has its own GPU
int32_t rls_comm_debug_busy_seconds(rls_comm* c, double* out) {
  if (!c || !out) return RLS_E_INVALID;
  for (int r = 0; r < c->n; ++r) {
    out[r] = c->pool ? c->pool->busy_s[r] : 0.0;
    if (c->pool) c->pool->busy_s[r] = 0.0;
  }
  return 0;
}

int32_t rls_comm_set_threads(rls_comm* c, int32_t on) {
  if (!c) return RLS_E_INVALID;
  if (!on) pool_stop(c->pool);
  c->use_threads = on ? 1 : 0;
  return 0;
}

int32_t rls_comm_create(int32_t nranks, const int32_t* devices, rls_ctx* const* ctxs, int32_t transport, rls_comm** out) {
  if (!out) return RLS_E_INVALID;
  *out = nullptr;
  if (nranks < 1 || nranks > MAX_RANKS || (!devices && !ctxs)) return RLS_E_INVALID;
  rls_comm* c = new rls_comm();
  c->n = nranks;
  c->own_ctx = ctxs == nullptr;
  if (const char* e = getenv("RLS_COMM_THREADS")) c->use_threads = atoi(e) != 0;
  for (int r = 0; r < nranks; ++r) {
    rls_ctx* x = nullptr;
    if (ctxs) {
      x = ctxs[r];
      if (!x || (devices && devices[r] != x->device)) {
        delete c;
        return RLS_E_INVALID;
      }
    } else {
      const int32_t st = rls_ctx_create(devices[r], &x);
      if (st != 0) {
        for (rls_ctx* y : c->ctx) rls_ctx_destroy(y);
        delete c;
        return st;
      }
    }
    c->ctx.push_back(x);
  }
  bool distinct = true;
  for (int r = 0; r < nranks; ++r)
    for (int t = 0; t < r; ++t) distinct = distinct && c->ctx[r]->device != c->ctx[t]->device;
  // Peer-access probe, whatever the transport: the direct transport stores into the peers' receive buffers, which needs
  // hipDeviceCanAccessPeer between every pair of distinct devices.  A pair without it is REPORTED (rls_comm_peer_access) and a
  // requested direct transport drops to RCCL when the ranks have a device each -- it never turns into a fault at the first
  // exchange.  (A query that fails is treated as "cannot".)
  c->peer.assign((size_t)nranks * nranks, 1);
  bool all_peers = true;
  for (int r = 0; r < nranks; ++r)
    for (int t = 0; t < nranks; ++t) {
      const int da = c->ctx[r]->device, db = c->ctx[t]->device;
      if (da == db) continue;
      int can = 0;
      const hipError_t e = hipDeviceCanAccessPeer(&can, da, db);
      if (e != hipSuccess) (void)hipGetLastError();
      c->peer[(size_t)r * nranks + t] = e != hipSuccess ? -1 : (can ? 1 : 0);
      all_peers = all_peers && e == hipSuccess && can;
    }
  c->requested = transport;
  if (transport == RLS_COMM_AUTO) transport = (distinct && nranks > 1) ? RLS_COMM_RCCL : RLS_COMM_DIRECT;
  if (transport == RLS_COMM_DIRECT && !all_peers && distinct) transport = RLS_COMM_RCCL;  // reported, not fatal
  c->transport = transport;
  int32_t st = 0;
  if (transport == RLS_COMM_RCCL) {
    if (!distinct) st = comm_fail(c, RLS_E_UNSUPPORTED, "rls_comm_create: the RCCL transport needs one distinct device per rank");
    if (st == 0) st = rccl_load(c);
    if (st == 0) {
      std::vector<int> devs;
      for (rls_ctx* x : c->ctx) devs.push_back(x->device);
      c->comms.resize(nranks);
      st = rccl_check(c, c->rccl.CommInitAll(c->comms.data(), nranks, devs.data()), "ncclCommInitAll");
      if (st != 0) c->comms.clear();
    }
  } else if (transport == RLS_COMM_DIRECT) {
    for (int q = 0; q < 2; ++q) c->recv[q].assign(nranks, nullptr);
    for (int q = 0; q < 2; ++q) c->pushed[q].assign(nranks, nullptr);
    for (int r = 0; r < nranks && st == 0; ++r) {
      hipError_t e = hipSetDevice(c->ctx[r]->device);
      for (int t = 0; t < nranks && e == hipSuccess; ++t) {
        const int da = c->ctx[r]->device, db = c->ctx[t]->device;
        if (da == db) continue;
        int can = 0;
        e = hipDeviceCanAccessPeer(&can, da, db);
        if (e == hipSuccess && !can) {
          st = comm_fail(c, RLS_E_UNSUPPORTED, "rls_comm_create: direct transport needs peer access between the ranks' devices");
          break;
        }
        if (e == hipSuccess) {
          e = hipDeviceEnablePeerAccess(db, 0);
          if (e == hipErrorPeerAccessAlreadyEnabled) {
            e = hipSuccess;
            (void)hipGetLastError();
          }
        }
      }
      for (int q = 0; q < 2; ++q)
        if (st == 0 && e == hipSuccess) e = hipEventCreateWithFlags(&c->pushed[q][r], hipEventDisableTiming);
      if (st == 0 && e != hipSuccess) st = comm_fail(c, (int32_t)e, hipGetErrorString(e));
    }
  } else {
    st = comm_fail(c, RLS_E_INVALID, "rls_comm_create: unknown transport");
  }
  if (st != 0) {
    rls_comm_destroy(c);
    return st;
  }
  *out = c;
  return 0;
}

int32_t rls_comm_destroy(rls_comm* c) {
  if (!c) return RLS_E_INVALID;
  pool_stop(c->pool);
  for (int r = 0; r < (int)c->ctx.size(); ++r) {
    hipSetDevice(c->ctx[r]->device);
    rls_stream_wait(c->ctx[r]->stream);
    for (int q = 0; q < 2; ++q)
      if (r < (int)c->recv[q].size() && c->recv[q][r]) hipFree(c->recv[q][r]);
    for (int q = 0; q < 2; ++q)
      if (r < (int)c->pushed[q].size() && c->pushed[q][r]) hipEventDestroy(c->pushed[q][r]);
  }
  for (nccl_comm_t k : c->comms)
    if (k && c->rccl.CommDestroy) c->rccl.CommDestroy(k);
  if (c->own_ctx)
    for (rls_ctx* x : c->ctx) rls_ctx_destroy(x);
  // the RCCL handle stays loaded: unloading a library with live device state is not safe
  delete c;
  return 0;
}

int32_t rls_comm_size(rls_comm* c) { return c ? c->n : RLS_E_INVALID; }
int32_t rls_comm_transport(rls_comm* c) { return c ? c->transport : RLS_E_INVALID; }

// the probe of rls_comm_create: out_matrix[r * nranks + t] (may be null) as described at rls_comm::peer; *out_requested = the
// transport asked for, the return value of rls_comm_transport is the one in use (they differ when AUTO resolved, or when a
// requested direct transport was dropped to RCCL because some pair of devices has no peer access)
int32_t rls_comm_peer_access(rls_comm* c, int32_t* out_matrix, int32_t* out_requested) {
  if (!c) return RLS_E_INVALID;
  if (out_matrix)
    for (size_t i = 0; i < c->peer.size(); ++i) out_matrix[i] = c->peer[i];
  if (out_requested) *out_requested = c->requested;
  return 0;
}

int32_t rls_comm_ctx(rls_comm* c, int32_t rank, rls_ctx** out) {
  if (!c || !out || rank < 0 || rank >= c->n) return RLS_E_INVALID;
  *out = c->ctx[rank];
  return 0;
}

int32_t rls_comm_sync(rls_comm* c) {
  if (!c) return RLS_E_INVALID;
  for (rls_ctx* x : c->ctx) RLS_TRY(rls_ctx_sync(x));
  return 0;
}

// rank_bufs[r]: device pointer on rank r's device, n elements of dtype; in place: every buffer ends up holding the
// sum over ranks.  Enqueued on the ranks' context streams (asynchronous).
int32_t rls_allreduce_sum(rls_comm* c, void* const* rank_bufs, int64_t n, int32_t dtype) {
  if (!c || !rank_bufs) return RLS_E_INVALID;
  if (!rls_dtype_ok(dtype) || n < 0) return comm_fail(c, RLS_E_INVALID, "rls_allreduce_sum: bad dtype or length");
  for (int r = 0; r < c->n; ++r)
    if (!rank_bufs[r]) return comm_fail(c, RLS_E_INVALID, "rls_allreduce_sum: null buffer");
  if (n == 0 || c->n == 1) return 0;
  const size_t nf = (size_t)n * (dtype == RLS_C32 ? 2 : 1);
  if (c->transport == RLS_COMM_DIRECT) return direct_allreduce(c, rank_bufs, nf);
  RLS_TRY(rccl_check(c, c->rccl.GroupStart(), "ncclGroupStart"));
  int32_t st = 0;
  for (int r = 0; r < c->n && st == 0; ++r)
    st = rccl_check(c, c->rccl.AllReduce(rank_bufs[r], rank_bufs[r], nf, /* ncclFloat32 */ 7, /* ncclSum */ 0, c->comms[r], c->ctx[r]->stream),
                    "ncclAllReduce");
  const int32_t st2 = rccl_check(c, c->rccl.GroupEnd(), "ncclGroupEnd");
  return st != 0 ? st : st2;
}

}  // extern "C"
