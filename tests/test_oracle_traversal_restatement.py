"""A second, independent restatement of the traversal state machine -- SimpleKNNCache
(simple_knn_cache.cuh:73-352: init, criteria, push, pop, fetch, transform) and the query loop
(query_layer.cu:39-97) -- as a lock-step thread emulation in Python, written from the cited
reference lines and sharing no code with oracle/ggnn_oracle.cpp or oracle/wave_model.hpp.  The
C++ oracle (which the HIP kernels are compared with bit for bit) must produce the same ids,
distances, distance-evaluation counts and pop counts.

Emulation model: the BLOCK threads of one workgroup run each statement in lock step, phases are
separated where the reference has a barrier; inside push's write phase every thread's shift
store precedes every thread's insert test (the only order under which the
reference is race-free).  Integer data: squared L2 distances are exact in float32.
"""
import numpy as np
import pytest

EMPTY_KEY = -1
EMPTY_DIST = np.float32(np.inf)


class PyCache:
    def __init__(self, BEST, SORTED, CACHE, BLOCK, xi, dist_fn):
        self.BEST, self.SORTED, self.CACHE, self.BLOCK = BEST, SORTED, CACHE, BLOCK
        self.s_cache = np.full(CACHE, EMPTY_KEY, np.int64)          # init(), :73-87
        self.s_dists = np.full(SORTED, EMPTY_DIST, np.float32)
        self.r_prioQ_head = BEST
        self.r0_visited_head = SORTED
        self.r_xi = np.float32(xi)
        self.dist_fn = dist_fn
        self.dist_calc_counter = 0

    def criteria(self):                                             # :121-124
        return np.float32(self.s_dists[self.BEST - 1] + self.r_xi)

    def push(self, key, dist):                                      # :126-213
        B, S, BLOCK = self.BEST, self.SORTED, self.BLOCK
        if (self.s_cache[:S] == key).any():                         # :131-146
            return
        head = self.r_prioQ_head
        head_in = head - B
        t = np.arange(BLOCK)
        idx = np.zeros(BLOCK, np.int64)
        r_cache = np.zeros(BLOCK, np.int64)
        r_dists = np.zeros(BLOCK, np.float32)
        active = np.zeros(BLOCK, bool)
        block_start = ((S + BLOCK - 1) // BLOCK) * BLOCK            # :161
        while True:
            if active.any():
                # shift (:165-174) -- all threads, then the insert test (:176-183)
                for l in np.nonzero(active)[0]:
                    if r_cache[l] != EMPTY_KEY:
                        nxt = B if idx[l] + 1 == S else idx[l] + 1
                        if nxt != B and nxt != head:
                            self.s_cache[nxt] = r_cache[l]
                            self.s_dists[nxt] = r_dists[l]
                ins = []
                for l in np.nonzero(active)[0]:
                    has_prev = idx[l] != 0 and idx[l] != head
                    prev = idx[l] - 1 if idx[l] != B else S - 1
                    if not has_prev or self.s_dists[prev] < dist:
                        ins.append(idx[l])
                for i in ins:
                    self.s_cache[i] = key
                    self.s_dists[i] = dist
            if block_start == 0:                                    # :186-187
                break
            block_start -= BLOCK                                    # :190-192
            idx = block_start + t
            active = idx < S
            ring = active & (idx >= B)                              # :198-202
            wrapped = np.where(idx + head_in < S, idx + head_in, idx + head_in - S + B)
            idx = np.where(ring, wrapped, idx)
            safe = np.where(active, idx, 0)
            r_cache = self.s_cache[safe]
            r_dists = self.s_dists[np.minimum(safe, S - 1)]
            active = active & (r_dists >= dist)                     # :208

    def pop(self):                                                  # :215-239
        head = self.r_prioQ_head
        key = int(self.s_cache[head])
        dist = self.s_dists[head]
        if key == EMPTY_KEY or dist >= self.criteria():
            return EMPTY_KEY
        v = self.r0_visited_head
        self.s_cache[v] = key
        self.r0_visited_head = self.SORTED if v + 1 >= self.CACHE else v + 1
        self.s_cache[head] = EMPTY_KEY
        self.s_dists[head] = EMPTY_DIST
        self.r_prioQ_head = self.BEST if head + 1 >= self.SORTED else head + 1
        return key

    def fetch(self, s_keys, translation, filter_known):             # :241-289
        if filter_known:
            for t in range(self.BLOCK):                             # :248-260, per-thread stride
                for i in range(t, self.CACHE, self.BLOCK):
                    n = self.s_cache[i]
                    if n == EMPTY_KEY:
                        if i >= self.SORTED:
                            break
                        continue
                    s_keys[s_keys == n] = EMPTY_KEY
        for k in range(len(s_keys)):                                # :267-283: ascending order
            other_n = int(s_keys[k])
            if other_n == EMPTY_KEY:
                continue
            other_m = int(translation[other_n]) if translation is not None else other_n
            self.dist_calc_counter += 1
            d = self.dist_fn(other_m)
            if d < self.criteria():
                self.push(other_n, d)

    def transform(self, tr):                                        # :297-333
        B, S = self.BEST, self.SORTED
        old_c, old_d = self.s_cache.copy(), self.s_dists.copy()
        for i in range(self.CACHE):
            if i < B:
                key = old_c[i]
                if key != EMPTY_KEY:
                    key = tr[key]
                self.s_cache[i] = key
                if i + B < S:
                    self.s_cache[i + B] = key
                    self.s_dists[i + B] = old_d[i]
            elif i < 2 * B and i < S:
                pass
            else:
                self.s_cache[i] = EMPTY_KEY
                if i < S:
                    self.s_dists[i] = EMPTY_DIST
        self.r_prioQ_head = B
        self.r0_visited_head = S


def bit_ceil(v):
    p = 1
    while p < v:
        p *= 2
    return p


def py_query(base, q, graph0, start, nn1_stats, KQuery, tau, max_iters, cosine=False):
    """query_kernels.cu:55-110 (sizing) + query_layer.cu:48-90"""
    KBuild = graph0.shape[1]
    required_sorted = ((KQuery + 1 + 16 + 31) // 32) * 32
    cache_size = max(256, required_sorted + 32, bit_ceil(max_iters))
    sorted_size = max(64 if cache_size < 512 else 32, required_sorted)
    tau = np.float32(tau)
    nmax = np.float32(nn1_stats[1])
    if cosine:
        xi = np.float32(nmax * tau)
    else:
        xi = np.float32(np.float32(np.float32(nmax * nmax) * tau) * tau)

    q64 = q.astype(np.float64)

    def dist_fn(m):
        if cosine:
            b = base[m].astype(np.float64)
            nn = np.float32((q64 * q64).sum()) * np.float32((b * b).sum())
            if not nn > 0:
                return np.float32(1.0)
            return np.float32(abs(np.float32(1.0) - np.float32((q64 * b).sum()) / np.float32(np.sqrt(np.float32(nn)))))
        d = base[m].astype(np.float64) - q64
        return np.float32((d * d).sum())

    c = PyCache(KQuery, sorted_size, cache_size, 32, xi, dist_fn)
    c.fetch(np.array(start, np.int64), None, False)
    pops = 0
    for _ in range(max_iters):
        if cosine:
            c.r_xi = min(xi, np.float32(c.s_dists[0] * tau))
        else:
            c.r_xi = min(xi, np.float32(np.float32(c.s_dists[0] * tau) * tau))
        a = c.pop()
        if a == EMPTY_KEY:
            break
        pops += 1
        for i in range(0, KBuild, 32):
            s_knn = np.full(32, EMPTY_KEY, np.int64)
            n = min(32, KBuild - i)
            s_knn[:n] = graph0[a, i:i + n]
            c.fetch(s_knn, None, True)
    return (c.s_cache[:KQuery].astype(np.int32), c.s_dists[:KQuery].copy(), c.dist_calc_counter,
            pops)


def _start_points(g):
    cfg = g["cfg"]
    return g["tr"][cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]


@pytest.mark.parametrize("K,tau,iters", [(10, 0.5, 400), (10, 0.9, 200), (40, 0.7, 64),
                                         (1, 0.3, 20), (100, 0.6, 300)])
def test_query_equals_python_restatement(orc, small_graph, K, tau, iters):
    """sorted 32 (cache 512), sorted 64 (cache 256), K = 40 (sorted 64: the ring wraps often),
    K = 1, K = 100 (sorted 128: four 32-thread chunks per push)"""
    g = small_graph
    base = g["base"]
    q = np.random.default_rng(K * 1000 + iters).integers(0, 256, (12, g["D"])).astype(np.float32)
    start = _start_points(g)
    ids, dists, nd, npop = orc.query(base, q, g["graph"][:g["N"]], start, g["stats"], K, tau, iters,
                                     counters=True)
    for i in range(q.shape[0]):
        p_ids, p_d, p_nd, p_pop = py_query(base, q[i], g["graph"][:g["N"]], start, g["stats"], K,
                                           tau, iters)
        assert np.array_equal(ids[i], p_ids), i
        assert dists[i].tobytes() == p_d.tobytes(), i
        assert (int(nd[i]), int(npop[i])) == (p_nd, p_pop), i


def test_query_on_tied_data_equals_python_restatement(orc):
    """few distinct coordinates: most pushes meet equal distances (quirk Q2) and the ring
    wrap of quirk Q1 is exercised with duplicates"""
    N, D, KB = 1500, 16, 24
    base = np.random.default_rng(1).integers(0, 3, (N, D)).astype(np.float32)
    cfg, graph, tr, sel, stats = orc.build(base, KB, 0.5, 1, rng=orc.make_rng(N, 5))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    q = np.random.default_rng(2).integers(0, 3, (10, D)).astype(np.float32)
    for K, tau, iters in ((10, 0.9, 200), (20, 1.5, 100)):
        ids, dists, nd, npop = orc.query(base, q, graph[:N], start, stats, K, tau, iters,
                                         counters=True)
        for i in range(q.shape[0]):
            p = py_query(base, q[i], graph[:N], start, stats, K, tau, iters)
            assert np.array_equal(ids[i], p[0]) and dists[i].tobytes() == p[1].tobytes(), (K, i)
            assert (int(nd[i]), int(npop[i])) == (p[2], p[3]), (K, i)


def test_cache_scripts_equal_python_restatement(orc):
    """random push / pop / xi / transform scripts against the oracle's literal cache (the entry
    point the wave-model test uses; its transform op applies the identity table), including tiny
    rings (BEST = 1, two queue slots) where quirk Q1 fires on nearly every wrapped push"""
    rs = np.random.default_rng(11)
    ident = np.arange(1000)
    for case in range(80):
        BEST = int(rs.integers(1, 20))
        SORTED = int(rs.integers(BEST + 1, BEST + 40))
        CACHE = SORTED + int(rs.integers(4, 64))
        xi = float(rs.integers(0, 50))
        ops, pops = [], []
        c = PyCache(BEST, SORTED, CACHE, 32, xi, None)
        for _ in range(int(rs.integers(20, 160))):
            r = rs.random()
            if r < 0.68:
                k, d = int(rs.integers(0, 200)), float(rs.integers(0, 60))
                ops.append(orc.op_push(k, d))
                c.push(k, np.float32(d))
                pops.append(-2)
            elif r < 0.94:
                ops.append(orc.op_pop())
                pops.append(c.pop())
            elif r < 0.98:
                x = float(rs.integers(0, 50))
                ops.append(orc.op_xi(x))
                c.r_xi = np.float32(x)
                pops.append(-2)
            else:
                ops.append(orc.op_transform())
                c.transform(ident)
                pops.append(-2)
        keys, dists, o_pops, heads = orc.cache_script(BEST, SORTED, CACHE, 32, xi, ops)
        assert np.array_equal(keys.astype(np.int64), c.s_cache), case
        assert dists.tobytes() == c.s_dists.tobytes(), case
        assert o_pops.tolist() == pops, case
        assert heads.tolist() == [c.r_prioQ_head, c.r0_visited_head], case


# ---- merge_layer.cu:40-158 (+ merge_layer.cuh:40-65) ---------------------------------------------
def py_merge_point(base, cfg, graph_all, tr_all, sel_all, nn1_stats, tau_build, layer_top,
                   layer_btm, n):
    KBuild, S, G = cfg.KBuild, cfg.S, cfg.G
    MAX_ITERATIONS, CACHE_SIZE, MIN_PRIOQ = 200, 256, 16
    SORTED_SIZE = max(64 if CACHE_SIZE < 512 else 32, ((KBuild + 1 + MIN_PRIOQ + 31) // 32) * 32)
    tau = np.float32(tau_build)
    mean = np.float32(nn1_stats[0])
    xi = np.float32(np.float32(np.float32(mean * mean) * tau) * tau)              # :74-76
    STs, Ns_off = list(cfg.STs_offsets), list(cfg.Ns_offsets)
    m = n if layer_btm == 0 else int(tr_all[STs[layer_btm] + n])                # :80
    p64 = base[m].astype(np.float64)

    def dist_fn(other_m):
        d = base[other_m].astype(np.float64) - p64
        return np.float32((d * d).sum())

    c = PyCache(KBuild + 1, SORTED_SIZE, CACHE_SIZE, 32, xi, dist_fn)
    # get_top_seg_offset, :40-61
    seg_btm = n // S
    if layer_btm == 0:
        off_pts = cfg.S0_off * (cfg.S0 + 1)
        seg_btm = n // (cfg.S0 + 1) if n < off_pts else cfg.S0_off + (n - off_pts) // cfg.S0
    powG = G
    for _ in range(1, layer_top - layer_btm):
        powG *= G
    s_offset = (seg_btm // powG) * S
    for i in range(0, S, 32):                                                     # :89-96
        s_knn = np.full(32, EMPTY_KEY, np.int64)
        for t in range(32):
            if i + t < S:
                s_knn[t] = s_offset + i + t
        c.fetch(s_knn, tr_all[STs[layer_top]:], False)
    layer = layer_top - 1
    while layer >= layer_btm:                                                     # :100-121
        c.transform(sel_all[STs[layer + 1]:])
        tr_l = None if layer == 0 else tr_all[STs[layer]:]
        if layer == layer_btm:
            c.fetch(np.array([n], np.int64), tr_l, False)
        for _ in range(MAX_ITERATIONS):
            a = c.pop()
            if a == EMPTY_KEY:
                break
            for j in range(0, KBuild, 32):
                s_knn = np.full(32, EMPTY_KEY, np.int64)
                cnt = min(32, KBuild - j)
                s_knn[:cnt] = graph_all[Ns_off[layer] + a, j:j + cnt]
                c.fetch(s_knn, tr_l, True)
        layer -= 1
    own = -1                                                                      # :123-136
    for k in range(KBuild):
        if c.s_cache[k] == n:
            own = k
    row = np.empty(KBuild, np.int32)
    for k in range(KBuild):                                                       # :138-145 (Q3)
        idx = c.s_cache[k + (1 if k >= own else 0)]
        row[k] = idx if idx != EMPTY_KEY else n
    nn1 = None
    if layer_btm == 0:                                                            # :147-157
        i = own + 1
        while True:
            d = c.s_dists[i]
            i += 1
            if not (d == 0.0 and i < c.BEST):
                break
        nn1 = np.float32(np.sqrt(np.float32(d)))
    return row, nn1


@pytest.mark.parametrize("top,btm", [(3, 0), (3, 2), (3, 1), (1, 0), (2, 1)])
def test_merge_rows_equal_python_restatement(orc, small_graph, top, btm):
    """sampled points of every merge the build and refine schedule launches (2->1 and 1->0 with
    the partially built graph below a full upper layer are the construction pairs, 3->l the
    refinement)"""
    g = small_graph
    cfg = g["cfg"]
    gb, nn1 = orc.merge(g["base"], cfg, g["graph"], g["tr"], g["sel"], g["stats"], 0.5, top, btm)
    Nb = cfg.Ns[btm]
    pts = sorted(set(np.random.default_rng(top * 10 + btm).integers(0, Nb, 14).tolist()
                     + [0, Nb - 1]))
    for n in pts:
        row, p_nn1 = py_merge_point(g["base"], cfg, g["graph"], g["tr"], g["sel"], g["stats"], 0.5,
                                    top, btm, n)
        assert np.array_equal(gb[n], row), (n, gb[n], row)
        if btm == 0:
            assert np.float32(nn1[n]).tobytes() == p_nn1.tobytes(), n


def test_merge_with_duplicate_points_equals_python_restatement(orc):
    """exact duplicates in the base: zero distances next to self (the nn1 loop of :147-157 skips
    them) and a point that may not find ITSELF among its first K entries (quirk Q3)"""
    N, D, KB = 1500, 8, 24
    rs = np.random.default_rng(4)
    base = rs.integers(0, 256, (N, D)).astype(np.float32)
    base[rs.integers(0, N, 400)] = base[rs.integers(0, 40, 400)]   # many copies of 40 rows
    cfg, graph, tr, sel, stats = orc.build(base, KB, 0.5, 0, rng=orc.make_rng(N, 9))
    gb, nn1 = orc.merge(base, cfg, graph, tr, sel, stats, 0.5, 3, 0)
    for n in rs.integers(0, N, 16).tolist():
        row, p_nn1 = py_merge_point(base, cfg, graph, tr, sel, stats, 0.5, 3, 0, n)
        assert np.array_equal(gb[n], row), n
        assert np.float32(nn1[n]).tobytes() == p_nn1.tobytes(), n


# ---- sym_query_layer.cu:39-145 + simple_knn_sym_cache.cuh:143-436 ---------------------------------
class PySymCache(PyCache):
    """the sym cache shares push with the plain cache (simple_knn_sym_cache.cuh:285-372 is the same
    text); pop compares with the NEAREST entry (:374-398, criteria_sym :280-283), the filter scans
    the whole cache (:403-414) and a candidate must also be close to the half point (:425)"""

    def __init__(self, BEST, SORTED, CACHE, xi, base, m):
        super().__init__(BEST, SORTED, CACHE, 32, xi, None)
        self.base = base
        self.q = base[m].astype(np.float32)
        self.half = None
        self.r_criteria_half = None

    def criteria_sym(self):
        return np.float32(self.s_dists[0] + self.r_xi)

    def dists(self, other_m):                                       # :213-278 (Euclidean)
        o = self.base[other_m].astype(np.float64)
        dq = self.q.astype(np.float64) - o
        dh = self.half.astype(np.float64) - o
        return np.float32((dq * dq).sum()), np.float32((dh * dh).sum())

    def init_start_point(self, other_n, translation):                # :143-197
        other_m = other_n if translation is None else int(translation[other_n])
        o = self.base[other_m].astype(np.float32)
        c = np.float32(0.5) - np.float32(0.1)
        self.half = (self.q + c * (o - self.q)).astype(np.float32)
        dq, dh = self.dists(other_m)
        self.r_criteria_half = np.float32(dh + self.r_xi)
        self.s_cache[:] = EMPTY_KEY
        self.s_dists[:] = EMPTY_DIST
        for i in (0, self.BEST):
            self.s_cache[i] = other_n
            self.s_dists[i] = dq
        self.r_prioQ_head = self.BEST
        self.r0_visited_head = self.SORTED

    def pop(self):
        head = self.r_prioQ_head
        key = int(self.s_cache[head])
        if key == EMPTY_KEY or self.s_dists[head] >= self.criteria_sym():
            return EMPTY_KEY
        v = self.r0_visited_head
        self.s_cache[v] = key
        self.r0_visited_head = self.SORTED if v + 1 >= self.CACHE else v + 1
        self.s_cache[head] = EMPTY_KEY
        self.s_dists[head] = EMPTY_DIST
        self.r_prioQ_head = self.BEST if head + 1 >= self.SORTED else head + 1
        return key

    def fetch_sym(self, s_keys, translation):
        for n in self.s_cache:
            if n != EMPTY_KEY:
                s_keys[s_keys == n] = EMPTY_KEY
        for k in range(len(s_keys)):
            other_n = int(s_keys[k])
            if other_n == EMPTY_KEY:
                continue
            other_m = other_n if translation is None else int(translation[other_n])
            dq, dh = self.dists(other_m)
            if dq < self.criteria_sym() and dh < self.r_criteria_half:
                self.push(other_n, dq)


def py_sym(base, KBuild, graph, translation, nn1_stats, tau_build, sym_buffer, sym_atomic, count):
    KF = KBuild // 2
    KL = KBuild - KF
    CACHE, MAX_PER_PATH = 128, 20
    sorted_size = max(64 if CACHE < 512 else 32, ((KF + 16 + 31) // 32) * 32)
    tau = np.float32(tau_build)
    mean = np.float32(nn1_stats[0])
    xi = np.float32(np.float32(np.float32(mean * mean) * tau) * tau)
    for n in range(count):
        m = n if translation is None else int(translation[n])
        c = PySymCache(KF, sorted_size, CACHE, xi, base, m)
        for k in range(KL):
            c.init_start_point(int(graph[n, k]), translation)
            found = False
            for _ in range(MAX_PER_PATH):
                anchor = c.pop()
                if anchor == EMPTY_KEY:
                    break
                for i in range(0, KBuild, 32):
                    s_knn = np.full(32, EMPTY_KEY, np.int64)
                    for t in range(32):
                        kk = i + t
                        if kk < KBuild:
                            s_knn[t] = graph[anchor, kk] if kk < KL else sym_buffer[anchor, kk - KL]
                    if (s_knn[:min(32, KBuild - i)] == n).any():
                        found = True
                        break
                    c.fetch_sym(s_knn, translation)
                if found:
                    break
            if not found:
                for i in range(KF):
                    other_n = int(c.s_cache[i])
                    if other_n == EMPTY_KEY:
                        break
                    pos = int(sym_atomic[other_n])
                    sym_atomic[other_n] += 1
                    if pos < KF:
                        sym_buffer[other_n, pos] = n
                        break


@pytest.mark.parametrize("KB,layer", [(24, 0), (20, 0), (24, 1)])
def test_sym_prefix_equals_python_restatement(orc, KB, layer):
    """the first points of one sym launch in the oracle's serial order (one point after the other,
    each seeing the inverse links its predecessors requested).  Values are multiples of 5: the
    half point q + 0.4 (start - q) is then exact in float32 (tests/test_gpu_build_parity.py)."""
    N, D = 1500, 32
    base = (np.random.default_rng(KB + layer).integers(0, 52, (N, D)) * 5).astype(np.float32)
    cfg, graph, tr, sel, stats = orc.build(base, KB, 0.5, 0, rng=orc.make_rng(N, 3))
    KF = KB // 2
    Nl = cfg.Ns[layer]
    g_layer = graph[cfg.Ns_offsets[layer]:cfg.Ns_offsets[layer] + Nl].copy()
    # the graph before sym: foreign links as merge wrote them
    tr_l = None if layer == 0 else tr[cfg.STs_offsets[layer]:cfg.STs_offsets[layer] + Nl].copy()
    count = min(Nl, 160)
    buf_a = np.full((Nl, KF), -1, np.int32)
    atom_a = np.zeros(Nl, np.uint32)
    orc.sym(base, KB, g_layer, tr_l, stats, 0.5, buf_a, atom_a, first_n=0, count=count)
    buf_b = np.full((Nl, KF), -1, np.int64)
    atom_b = np.zeros(Nl, np.int64)
    py_sym(base, KB, g_layer, tr_l, stats, 0.5, buf_b, atom_b, count)
    assert np.array_equal(atom_a.astype(np.int64), atom_b)
    assert np.array_equal(buf_a.astype(np.int64), buf_b)
    assert atom_b.sum() > 0, "no inverse link was requested: the case tests nothing"


def test_cosine_query_equals_python_restatement(orc):
    """Cosine measure (distance.cuh:139-158: |1 - q.b / sqrt(|q|^2 |b|^2)|, xi = nn1_max * tau,
    query_layer.cu:48-61).  Small integer coordinates: dot products and norms are exact, the few
    float32 operations after them are correctly rounded on both sides."""
    N, D, KB = 2000, 24, 24
    base = np.random.default_rng(31).integers(1, 16, (N, D)).astype(np.float32)
    cfg, graph, tr, sel, stats = orc.build(base, KB, 0.5, 1, measure=orc.COSINE,
                                           rng=orc.make_rng(N, 6))
    start = tr[cfg.STs_offsets[3]:cfg.STs_offsets[3] + cfg.Ns[3]]
    q = np.random.default_rng(32).integers(1, 16, (10, D)).astype(np.float32)
    ids, dists, nd, npop = orc.query(base, q, graph[:N], start, stats, 10, 0.8, 200,
                                     measure=orc.COSINE, counters=True)
    for i in range(q.shape[0]):
        p = py_query(base, q[i], graph[:N], start, stats, 10, 0.8, 200, cosine=True)
        assert np.array_equal(ids[i], p[0]), i
        assert dists[i].tobytes() == p[1].tobytes(), i
        assert (int(nd[i]), int(npop[i])) == (p[2], p[3]), i
