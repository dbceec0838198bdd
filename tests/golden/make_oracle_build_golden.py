"""Regression fixture for the ORACLE's build (not a pin to the reference: the reference has no
runnable build here).  Seeded multiples-of-5 data as in
tests/test_gpu_build_parity.py; stores SHA-256 digests of what `orc.build` returns, so that an
accidental change of the oracle -- the anchor every GPU parity test compares with -- is caught by
the CPU suite.  Regenerate deliberately with:  python tests/golden/make_oracle_build_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

CASES = [dict(N=3000, D=64, K=24, tau=0.5, refine=1, dtype="float32", seed=101, rng_seed=7),
         dict(N=2500, D=32, K=20, tau=0.6, refine=2, dtype="uint8", seed=202, rng_seed=9)]


def digest(case):
    v = np.random.default_rng(case["seed"]).integers(0, 52, (case["N"], case["D"])) * 5
    base = v.astype(case["dtype"])
    rng = orc.make_rng(case["N"], case["rng_seed"])
    cfg, graph, tr, sel, stats = orc.build(base, case["K"], case["tau"], case["refine"], rng=rng)
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    return {"graph": h(graph), "translation": h(tr), "selection": h(sel), "nn1_stats": h(stats),
            "G": int(cfg.G), "N_all": int(cfg.N_all)}


if __name__ == "__main__":
    out = [dict(case=c, digest=digest(c)) for c in CASES]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_build.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
