"""Host-side arithmetic of the float bf kernels' work split (ggnn_amd/csrc/bf_mfma.hip,
launch_bf_query_mfma + "Work of this workgroup" in bf_mfma_kernel), restated in Python: the
(query block, unit) sequence is cut into equal ranges, a range that crosses a query-block boundary
is processed in segments, and every segment writes the part `block - first block of its query
block`.  Checked for many shapes: every (query block, unit) pair is covered exactly once, parts of
a query block are distinct and below the number of part slots the re-rank kernel is given."""
import itertools

import pytest

TILE = 32


def split(nq, n_base, resident, unit_tiles, slices_hook=0):
    qblocks = (nq + 127) // 128
    unit_rows = unit_tiles * TILE
    units_per_q = (n_base + unit_rows - 1) // unit_rows
    total = qblocks * units_per_q
    per = (total + resident - 1) // resident
    per = max(per, (units_per_q + 31) // 32)
    if slices_hook:
        per = (units_per_q + slices_hook - 1) // slices_hook
    per = max(1, per)
    nblocks = (total + per - 1) // per
    parts = (units_per_q + per - 1) // per + 1
    return qblocks, units_per_q, total, per, nblocks, parts, unit_rows


def segments(block, units_per_q, total, per, n_base, unit_rows):
    work, end = block * per, min(total, block * per + per)
    while work < end:
        qb = work // units_per_q
        q_first = qb * units_per_q
        t0 = work - q_first
        cnt = min(units_per_q - t0, end - work)
        yield qb, t0, cnt, block - q_first // per, t0 * unit_rows, min(n_base, (t0 + cnt) * unit_rows)
        work += cnt


SHAPES = [(10_000, 1_000_000), (256, 4096), (257, 4500), (110_000, 4100), (100_000, 1_000_000),
          (2_000, 1_000_000), (1_000_000, 5_000), (300, 123_457), (40_000, 4100)]


@pytest.mark.parametrize("nq,n_base", SHAPES)
@pytest.mark.parametrize("resident,unit_tiles", [(768, 1), (512, 2), (512, 3), (256, 4)])
@pytest.mark.parametrize("slices_hook", [0, 1, 7])
def test_every_unit_once_and_parts_fit(nq, n_base, resident, unit_tiles, slices_hook):
    qblocks, upq, total, per, nblocks, parts, unit_rows = split(nq, n_base, resident, unit_tiles,
                                                              slices_hook)
    assert nblocks * per >= total > (nblocks - 1) * per
    if not slices_hook:
        assert nblocks <= max(resident, 32 * qblocks)
    covered = {}
    parts_of = {}
    rows_of = {}
    # (bounded work: sample the blocks of big launches around query-block boundaries)
    blocks = range(nblocks) if nblocks <= 4096 else sorted(
        set(itertools.chain(range(64), range(nblocks - 64, nblocks),
                            *(range(max(0, b - 2), min(nblocks, b + 3))
                              for b in range(0, nblocks, max(1, nblocks // 257))))))
    for b in blocks:
        for qb, t0, cnt, part, begin, end in segments(b, upq, total, per, n_base, unit_rows):
            assert 0 <= part < parts
            assert begin < end <= n_base and begin == t0 * unit_rows
            assert (qb, part) not in parts_of, "two segments write the same part"
            parts_of[(qb, part)] = b
            for u in {t0, t0 + cnt - 1}:
                assert (qb, u) not in covered
                covered[(qb, u)] = b
            rows_of[qb] = rows_of.get(qb, 0) + (end - begin)
    if nblocks <= 4096:
        # all blocks visited: the segments of a query block tile its rows exactly
        assert set(rows_of) == set(range(qblocks))
        assert all(r == n_base for r in rows_of.values())
