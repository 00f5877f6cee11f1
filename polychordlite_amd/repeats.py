"""Independent repeats on ONE GPU.

A single run keeps about one wavefront per SIMD busy (B <= 1024 chains per nursery, the contraction on one CU), so
independent repeats of the same problem -- the runs BASELINE's repeat-sharded mode spreads over GPUs --
also overlap on one: R host threads, each driving its own engine instance on its own HIP stream (the C entry point
releases nothing Python-side: ctypes drops the GIL for the duration of the call).  The repeats are merged like the
multi-GPU ones (merge.evidence_replay over the union of their death records): sigma(logZ) falls like 1/sqrt(R).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ctypes_api as api
from .merge import evidence_replay, lived_records


def run_repeats(settings, like, prior, seeds, max_in_flight=4):
    """Run one nested-sampling run per seed, up to `max_in_flight` at a time on the current device, and merge them.

    Returns (merged, runs): merged = {"n_runs", "logZ", "logZerr", "records", "nlike", "t_runs_s", "t_merge_s"}; runs = the per-seed result
    dicts of `_ctypes_api.run` (zero-copy views of the engine's pinned buffers)."""
    seeds = list(seeds)
    if not seeds:
        raise ValueError("run_repeats needs at least one seed")
    copies = []
    for sd in seeds:                                  # one settings block per repeat: the call reads it while it runs
        s = api.Settings()
        C.memmove(C.byref(s), C.byref(settings), C.sizeof(settings))
        s.seed = int(sd)
        copies.append(s)
    import time
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max(1, min(int(max_in_flight), len(seeds)))) as ex:
        runs = list(ex.map(lambda s: api.run(s, like, prior), copies))
    t1 = time.perf_counter()
    rec = [lived_records(r) for r in runs]
    logL = np.concatenate([a for a, _ in rec]); entry = np.concatenate([b for _, b in rec])
    lz, var = evidence_replay(logL, entry)
    merged = {"n_runs": len(runs), "logZ": lz, "logZerr": float(np.sqrt(abs(var))), "records": int(logL.size),
              "nlike": int(sum(r["nlike"] for r in runs)), "t_runs_s": t1 - t0, "t_merge_s": time.perf_counter() - t1}
    return merged, runs
