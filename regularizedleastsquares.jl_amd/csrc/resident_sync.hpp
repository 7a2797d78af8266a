// In-launch hand-off primitives of the resident kernels (normal.hip, gramk.hip): write-through stores and L1-bypassing
// loads, the sync block with its sharded arrival counters, and the bounded grid / group barriers.
#pragma once
#include "rls_common.hpp"

// Write-through (sc1) accesses for data handed to other workgroups INSIDE a launch (the resident kernels below):
// an sc1 store leaves the XCD's L2 for the memory side at once (no release fence needed), an sc1 load bypasses
// the CU's L1, which another CU's stores never refresh.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ static inline __amdgpu_buffer_rsrc_t sc1_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xffffffff, 0x00020000);
}
__device__ static inline f4 sc1_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));  // aux 16 = sc1
}
template <typename E>
__device__ static inline void sc1_store_elem(E* p, E v) {
  if constexpr (sizeof(E) == 8)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  else
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}


template <typename E>
__device__ static inline E sc1_load_elem(const E* p) {
  if constexpr (sizeof(E) == 8)
    return __builtin_bit_cast(E, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
  else
    return __builtin_bit_cast(E, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT));
}


struct resident_sync {
  unsigned cnt[8 * 32];  // grid arrival counter (zeroed before every launch): 8 shards a 128-byte line apart
  unsigned gcnt[8 * 32]; // group arrival counters of the two-level exchange: one word per group, a line apart
  unsigned fail;         // some workgroup gave up waiting (last launch)
  unsigned completed;    // workgroup 0 passed the last barrier and wrote the state back (last launch)
  // ---- everything above is zeroed ahead of every launch (rls_resident_sync_clear_bytes); what follows is STICKY ----
  unsigned failed;       // launches of this plan that were no-ops because workgroup 0 gave up: only workgroup 0 writes the
                         // state back, so "workgroup 0 timed out" is exactly "the launch changed nothing".  Zeroed at plan
                         // creation and by the HOST once a status call has seen it (resident_lost: it re-runs what the current
                         // solve is missing and retires the plan from the resident kernels); init! does NOT clear it, so a loss
                         // nobody asked about is still reported -- by the next status call, of whichever solve.
  unsigned srv_n, srv_mb;  // server mode: the command workgroup 0 relays to the grid (n_steps or RLS_SRV_EXIT, mailbox sequence)
  unsigned pad[27];
};
static_assert(sizeof(resident_sync) == (2 * 8 * 32 + 32) * sizeof(unsigned), "resident_sync layout");
// behind the sync block (same allocation): the group-partial vectors of the two-level exchange, [2 parities][8 groups][N]
constexpr int RES_GROUPS = 8;

// a workgroup gave up waiting.  Workgroup 0 is the only one that writes x, r, p and the scalars back, so its giving up
// is what makes the launch a no-op: it counts the lost launch in the sticky word and, inside an ADMM plan, poisons the
// plan's `done` flag (value 2) so that the z / u kernels queued behind this cg! do not consume a stale x.
__device__ static inline void resident_give_up(resident_sync* sync, int* poison) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(&sync->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0) {
      __hip_atomic_fetch_add(&sync->failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (poison) __hip_atomic_store(poison, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}


// arrive + wait on the grid counter.  Precondition: this workgroup's handed-off stores are sc1, drained by every storing
// wave, and a workgroup barrier lies between those drains and this call.  Arrival = one returning-free atomic add to one
// of 8 counter shards (a 128-byte line each), waiting = lanes 0..7 of one wave re-reading the 8 shards until their sum
// reaches nwg * epoch.  (Round 2 also carried a variant with one flag word per workgroup; measured slower -- 18.3 vs 15.9 us
// per iteration: 256 pollers each pulling 8 lines that 32 writers share -- and removed in round 3.)
#ifndef RLS_POLL_SLEEP
#define RLS_POLL_SLEEP 1
#endif
// (`tid`: the caller's copy of threadIdx.x -- a kernel short of registers passes a per-iteration opaque copy so that the poll
// address is derived where it is used instead of being carried across its loop)
__device__ static inline bool grid_arrive_wait_t(unsigned* cnt, unsigned epoch, unsigned nwg, unsigned spin_limit, int* lds_flag,
                                                 const int tid) {
  if (tid < 64) {
    int ok = 0;
    const unsigned target = nwg * epoch;
    if (tid == 0) __hip_atomic_fetch_add(cnt + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ONE poll in flight, a short sleep between polls: the pollers of 256 workgroups share the memory-side path
    // with the arrivals they are waiting for.  Measured at the headline shape (us per iteration, one run): sleep 1
    // 14.3, sleep 8 14.7, sleep 32 15.4; two polls in flight (the next requested before the previous is examined) 16.2.
    for (unsigned spins = 0; spins < spin_limit; ++spins) {
      unsigned c = __hip_atomic_load(cnt + (tid & 7) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, true);   // lanes 0..7 hold the 8 shards:
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, true);   // butterfly inside the group of 8
      c += (unsigned)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true);
      if (__builtin_amdgcn_readfirstlane((int)c) >= (int)target) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(RLS_POLL_SLEEP);
    }
    if (tid == 0) *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}
__device__ static inline bool grid_arrive_wait(unsigned* cnt, unsigned epoch, unsigned nwg, unsigned spin_limit, int* lds_flag) {
  return grid_arrive_wait_t(cnt, epoch, nwg, spin_limit, lds_flag, (int)threadIdx.x);
}
// the same on ONE word: the members of a group (RES_GROUPS groups, workgroups with equal blockIdx % RES_GROUPS -- under the
// observed round-robin dispatch one XCD each, which only makes it faster) wait for each other
__device__ static inline bool group_arrive_wait(unsigned* word, unsigned target, unsigned spin_limit, int* lds_flag) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    int ok = 0;
    if (tid == 0) __hip_atomic_fetch_add(word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spins = 0; spins < spin_limit; ++spins) {
      const unsigned c = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__builtin_amdgcn_readfirstlane((int)c) >= (int)target) {
        ok = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(RLS_POLL_SLEEP);
    }
    if (tid == 0) *lds_flag = ok;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// ---- server mode (rls_cg_start::srv_ctl, rls_srv_args): a resident kernel that has finished a step call LISTENS for the next one ----
// Workgroup 0 polls the control block in pinned host memory -- [0] command sequence, [1] n_steps, [2] mailbox sequence; it writes
// [16] leaving, [17] exited (1 = left idle or on EXIT, 2 = a wait ran out) -- while the rest of the grid waits at a grid barrier
// whose bound (spin_limit polls) lies far beyond the idle time.  Returns the command's n_steps, or RLS_SRV_EXIT when the grid is
// to leave (then the control block already says why); srv_seq advances to the command now being served, mb.seq to its mailbox
// sequence number.  Called by every thread of every workgroup; the caller's write-back of the previous command lies before it.
// A kernel also leaves once it has served RLS_SRV_MAX_COMMANDS, command posted or not (a caller iterating for seconds must not
// turn into one kernel that runs for seconds: ~40 ms of one-iterate calls): the host finds `exited` instead of its status and
// re-issues the command with a launch, as for a kernel that left idle.
constexpr unsigned RLS_SRV_MAX_COMMANDS = 2048u;
// the head of the control block {command sequence, n_steps, mailbox sequence, -} in ONE 16-byte read at system scope: the block lives
// in pinned host memory, every read of it is a round trip over the link (1.5-2 us), and the three words used to be three dependent
// reads.  The host writes the payload, a release fence, then the sequence word (server_command, solvers.hip): a read that shows the new
// sequence shows the payload of that command (the 16 bytes are one aligned piece of one cache line).
struct srv_head {
  unsigned seq, n, mbseq, pad;
};
__device__ static inline srv_head srv_read_head(const unsigned* ctl) {
  typedef unsigned u4h __attribute__((ext_vector_type(4)));
  u4h w;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(ctl) : "memory");
  return srv_head{w.x, w.y, w.z, w.w};
}
__device__ static inline unsigned resident_listen(unsigned* ctl, unsigned& srv_seq, unsigned idle_us, resident_sync* sync, unsigned& epoch,
                                                  unsigned nwg, unsigned spin_limit, int* lds_flag, rls_mailbox_slot& mb,
                                                  unsigned served) {
  const int tid = threadIdx.x;
  if (blockIdx.x == 0 && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the write-back above is out before anything else is announced)
    const unsigned long long t0 = wall_clock64(), idle = (unsigned long long)idle_us * 100ull;  // 100 MHz
    unsigned n = RLS_SRV_EXIT;
    srv_head hd{srv_seq, 0u, 0u, 0u};
    for (; served < RLS_SRV_MAX_COMMANDS;) {
      hd = srv_read_head(ctl);
      if (hd.seq != srv_seq) break;
      if (wall_clock64() - t0 > idle) {
        // leave -- unless a command slips in: "leaving" goes out, THEN the sequence word is read once more (the host posts
        // its command and THEN reads "exited": one of the two sees the other)
        __hip_atomic_store(ctl + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hd = srv_read_head(ctl);
        if (hd.seq != srv_seq) __hip_atomic_store(ctl + 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    if (hd.seq != srv_seq) n = hd.n;
    __hip_atomic_store(&sync->srv_n, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sync->srv_mb, hd.mbseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (!grid_arrive_wait(sync->cnt, ++epoch, nwg, spin_limit, lds_flag)) {
    resident_give_up(sync, nullptr);
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(ctl + 17, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return RLS_SRV_EXIT;
  }
  const unsigned cmd = __hip_atomic_load(&sync->srv_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (cmd == RLS_SRV_EXIT) {  // uniform
    if (blockIdx.x == 0 && tid == 0) __hip_atomic_store(ctl + 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return RLS_SRV_EXIT;
  }
  mb.seq = __hip_atomic_load(&sync->srv_mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  srv_seq += 1;
  return cmd;
}
