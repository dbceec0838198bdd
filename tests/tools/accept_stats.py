"""Test-side tool (uses the oracle): fraction of distance evaluations that pass `d < criteria()`.

    python tests/tools/accept_stats.py            # CPU only, oracle-built graph

Measured with the oracle's statistics counters (orc_accept_total / orc_eval_total):
query on a GPU-built 1M graph 15.5 % accepted (tau 0.9, 200 iterations), whole build 8.6 %.
This is the measurement behind the exact pre-screen.
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from oracle import oracle as orc  # noqa: E402


def main(n=20000, d=128, k=24):
    rng = np.random.default_rng(1)
    mix = rng.normal(size=(16, d)) * 10.0
    base = np.clip(np.rint(128 + rng.normal(size=(n, 16)) @ mix), 0, 255).astype(np.float32)
    query = np.clip(np.rint(128 + rng.normal(size=(500, 16)) @ mix), 0, 255).astype(np.float32)
    lib = orc.lib()
    lib.orc_accept_total.restype = C.c_uint64
    lib.orc_eval_total.restype = C.c_uint64
    orc.set_fast_distance(True)
    lib.orc_accept_total(1), lib.orc_eval_total(1)
    cfg, graph, tr, sel, stats = orc.build(base, k, 0.5, 2, rng=orc.make_rng(n, 5))
    print(f"build: {lib.orc_accept_total(0) / lib.orc_eval_total(0):.3f} of "
          f"{lib.orc_eval_total(0)} evaluations accepted")
    lib.orc_accept_total(1), lib.orc_eval_total(1)
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    orc.query(base, query, graph[:n], start, stats, 10, 0.9, 200)
    print(f"query: {lib.orc_accept_total(0) / lib.orc_eval_total(0):.3f} of "
          f"{lib.orc_eval_total(0)} evaluations accepted")
    orc.set_fast_distance(False)


if __name__ == "__main__":
    main()
