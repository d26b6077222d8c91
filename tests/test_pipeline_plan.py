"""The rollout-unit plan of a pipelined run (slotformer_amd.pipeline.unit_sizes_for): a pure function of the batch counts, checked on the CPU.
The pipeline itself (streams, graphs, bit-identity of every plan with the serial calls) is tests/test_pipeline_gpu.py."""
import pytest

from slotformer_amd.pipeline import unit_sizes_for

C2_ROWS = 32 * 7 * 6      # token rows of a C2 batch (32 videos x 7 slots x 6 frames): 6 batches fill one round of row tiles
C5_ROWS = 64 * 6 * 8      # a C5 batch is more than a third of a round: units never grow


@pytest.mark.parametrize('n', list(range(0, 60)) + [100, 101, 203])
def test_sizes_cover_the_run(n):
    for rows in (C2_ROWS, C5_ROWS, 84):
        for G in (1, 2, 4, 7):
            for ramp in (False, True):
                sizes, n_tail = unit_sizes_for(n, G, rows, ramp=ramp)
                assert sum(sizes) == n and all(s >= 1 for s in sizes) and 0 <= n_tail <= len(sizes)
                if not ramp:
                    assert all(s * rows <= 8192 or s <= G for s in sizes)   # a unit grows beyond the group only within one round of row tiles
                    odd = [s for s in sizes if s != G]
                    assert len(odd) <= 2                                     # at most two units of another size (two unit objects per size)


def test_c2_plans():
    plan = lambda n, **k: unit_sizes_for(n, 4, C2_ROWS, **k)[0]  # noqa: E731
    assert plan(20) == [4, 4, 6, 6] and plan(21) == [4, 4, 4, 4, 5] and plan(22) == [4, 4, 4, 5, 5] and plan(23) == [4, 4, 4, 5, 6]
    assert plan(24) == [4, 4, 4, 6, 6] and plan(12) == [4, 4, 4] and plan(13) == [4, 4, 5] and plan(16) == [4, 6, 6]
    assert plan(100)[-2:] == [6, 6] and set(plan(100)[:-2]) == {4}
    # short runs and the opt-out keep the plain plan: full units + the remainder
    assert plan(11) == [4, 4, 3] and plan(5) == [4, 1] and plan(3) == [3] and plan(0) == []
    assert plan(21, spread=False) == [4, 4, 4, 4, 4, 1]
    # units that cannot grow (C5: a batch is 3072 rows) keep the remainder as a unit of its own
    assert unit_sizes_for(21, 4, C5_ROWS)[0] == [4, 4, 4, 4, 4, 1]
    # the ramp (off by default): the last batches in ever smaller units, which replace at most one full unit
    assert unit_sizes_for(20, 4, C2_ROWS, ramp=True) == ([4, 4, 4, 4, 2, 1, 1], 3)
