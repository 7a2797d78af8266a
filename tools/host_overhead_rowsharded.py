"""host-side cost per iteration of the single-process row-sharded hosts (rls_cgnr/fista_step_rowsharded): 1-8 ranks, tiny
shards.  On this one-GPU box every rank's kernels land on the SAME device, whose dispatcher takes ~1.7 us per dependent
tiny kernel, so the wall clock grows with the rank count whatever the host does (8 ranks = 40 kernels per iteration);
what a node with one GPU per rank would see as host time is the time ONE rank's worker spends enqueueing its share, which
the library counts (rls_comm_debug_busy_seconds).  VERDICT r2 item 1b: <= 30 us of host time per iteration for 8 ranks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import rls_amd as rls

rows, N, its = 64, 256, 2000
rng = np.random.default_rng(3)
for nshards in (1, 2, 4, 8):
  shards = [np.asfortranarray((rng.standard_normal((rows, N)) + 1j * rng.standard_normal((rows, N))).astype(np.complex64)) for _ in range(nshards)]
  parts = [rng.standard_normal(rows).astype(np.complex64) for _ in range(nshards)]
  for name, make in (("CGNR", lambda th: rls.CommRowShardedCGNR(rls, shards, transport=2, iterations=10 ** 9, relTol=0.0)),
                     ("FISTA", lambda th: rls.CommRowShardedFISTA(rls, shards, reg=rls.L1Regularization(1e-3), transport=2, rho=1e-3,
                                                                  iterations=10 ** 9, relTol=0.0, threads=th))):
      for threads in (True, False):
          s = make(threads)
          s.lib.rls_comm_set_threads(s.comm, 1 if threads else 0)
          s.init(parts)
          s.step(50)
          s.sync()
          best = 1e9
          busy = (C.c_double * nshards)()
          s.lib.rls_comm_debug_busy_seconds(s.comm, busy)
          for _ in range(3):
              t0 = time.perf_counter()
              s.step(its)
              t1 = time.perf_counter()
              s.sync()
              t2 = time.perf_counter()
              best = min(best, (t1 - t0) / its)
          s.lib.rls_comm_debug_busy_seconds(s.comm, busy)
          per_rank = max(busy) / (3 * its) * 1e6 if threads and nshards > 1 else float("nan")
          print(f"{name:5s} {nshards} ranks, worker threads {'on ' if threads else 'off'}: wall {best * 1e6:7.1f} us per iteration "
                f"(incl. drain {1e6 * (t2 - t0) / its:7.1f}); busiest rank's worker spent {per_rank:6.1f} us per iteration enqueueing", flush=True)
          s.close()
