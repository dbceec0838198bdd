"""Validates the kernel design on CPU: the wave64 "one sorted entry per lane, logical order"
model (oracle/wave_model.hpp, mirrored 1:1 by SortedList in ggnn_amd/csrc/traversal.hpp) must
reproduce the literal block-lockstep emulation of SimpleKNNCache, including the ring wrap."""
import numpy as np


def test_wave_model_equals_literal_emulation(orc):
    r = np.random.default_rng(5)
    P, POP, XI, TR = orc.op_push, orc.op_pop, orc.op_xi, orc.op_transform
    wraps = 0
    for trial in range(400):
        SORTED = int(r.choice([32, 64, 128]))
        BEST = int(r.integers(1, SORTED - 3))
        CACHE = int(r.choice([SORTED + 32, 256, 512]))
        ops = [XI(float(r.choice([0.5, 5.0, 1e9])))]
        for _ in range(int(r.integers(10, 500))):
            x = r.random()
            if x < 0.55:
                ops.append(P(int(r.integers(0, 300)), float(r.integers(0, 60))))
            elif x < 0.95:
                ops.append(POP())
            elif x < 0.985:
                ops.append(XI(float(r.choice([0.0, 1.0, 10.0, 1e9]))))
            else:
                ops.append(TR())
        a = orc.cache_script(BEST, SORTED, CACHE, 32, 0.0, ops)
        b = orc.wave_model_script(BEST, SORTED, CACHE, 0.0, ops)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (trial, BEST, SORTED, CACHE)
        wraps += int(a[3][0] != BEST)
    assert wraps > 50  # the ring was rotated in many of the trials


def test_wave_model_q1_trace(orc):
    P, POP, XI = orc.op_push, orc.op_pop, orc.op_xi
    ops = [XI(1e9), P(101, 1.0), P(102, 2.0), POP(), POP(), P(10, 10.0), P(20, 20.0),
           P(30, 30.0), P(15, 15.0)]
    k, d, _, h = orc.wave_model_script(2, 6, 16, 0.0, ops)
    assert list(k[2:6]) == [30, 30, 10, 15] and h[0] == 4
