"""Cosine fuzz (run on the GPU box): bf_query must contain the float64 top K up to rounding of
near-ties for random shapes / element types; the traversal must stay finite and reasonably
accurate (unstructured Gaussian rows are hard: recall 0.8-0.9 at tau 1.0 / 400 is expected)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import ggnn_amd as ggnn
ggnn.set_log_level(-1)
rng = np.random.default_rng(7); bad = 0
for case in range(40):
    D = int(rng.choice([3, 30, 64, 100, 128, 200, 384, 960])); N = int(rng.integers(5000, 20000)); Nq = int(rng.choice([7, 256, 300])); K = int(rng.choice([1, 10, 50]))
    dtype = rng.choice(["f32", "u8"])
    base = (rng.normal(size=(N, D)) * 3 + rng.normal(size=(1, D)) * 5).astype(np.float32) if dtype == "f32" else rng.integers(0, 256, (N, D)).astype(np.uint8)
    q = base[rng.integers(0, N, Nq)].astype(np.float32) + (rng.normal(size=(Nq, D)).astype(np.float32) if dtype == "f32" else 0)
    q = q.astype(base.dtype) if dtype == "f32" else np.clip(q + rng.integers(0, 9, (Nq, D)), 0, 255).astype(np.uint8)
    eng = ggnn.GGNN(); eng.set_base(base); eng.build(24, 0.5, 1, ggnn.DistanceMeasure.Cosine)
    gt, gd = eng.bf_query(q, K, ggnn.DistanceMeasure.Cosine)
    b64, q64 = base.astype(np.float64), q.astype(np.float64)
    d = np.abs(1 - (q64 @ b64.T) / np.maximum(np.linalg.norm(q64, axis=1)[:, None] * np.linalg.norm(b64, axis=1)[None], 1e-300))
    kth = np.sort(d, 1)[:, K - 1]
    got = np.take_along_axis(d, gt.numpy().astype(np.int64), 1)
    ok_bf = bool((got.max(1) <= kth + 2e-6).all())
    ids, dd = eng.query(q, min(K, 10), 1.0, 400, ggnn.DistanceMeasure.Cosine)
    rec = np.mean([len(set(a[:min(K,10)]) & set(b[:min(K,10)])) / min(K, 10) for a, b in zip(ids.numpy(), gt.numpy())])
    ok = ok_bf and rec > 0.7 and np.isfinite(dd.numpy()).all()
    bad += 0 if ok else 1
    print("ok" if ok else "BAD", dtype, N, D, Nq, K, ok_bf, round(rec, 3), flush=True)
print("failures:", bad)
