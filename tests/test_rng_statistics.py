"""The random streams as the kernels consume them (Philox4x32 at seven rounds, low halves hashed from the block's own words -
oracle/rt_oracle.hpp `Stream`, mirrored bit for bit by the device): statistical checks on the draw words themselves, beside the
known-answer vectors of tests/test_oracle_golden.py and the converged-image comparisons.  What a path tracer needs of them:
uniform draws, no dependence between the streams of neighbouring pixels / samples / bounces, between the slots of one stream,
or between the generated high half of a word and its hashed low half.

Every bound below is five standard deviations of the statistic under the null hypothesis (a false alarm in 10^6 runs or so);
the seeds are fixed, so the test is deterministic."""
import numpy as np
import pytest
from scipy import stats

N_STREAMS = 1 << 16
SLOTS = 8


@pytest.fixture(scope="module")
def words(oracle):
    """[stream, slot] draw words of N_STREAMS scatter streams: consecutive pixels x 4 samples x 4 bounces."""
    out = np.zeros((N_STREAMS, SLOTS), dtype=np.uint64)
    k = 0
    for pixel in range(N_STREAMS // 16):
        for sample in range(4):
            for bounce in range(4):
                out[k] = oracle.stream_words(2024, 1000 + pixel, sample, bounce, 2, SLOTS)
                k += 1
    return out


def _u01(w):
    return (w >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def test_draws_are_uniform(words):
    for slot in range(SLOTS):
        p = stats.kstest(_u01(words[:, slot]), "uniform").pvalue
        assert p > 1e-6, (slot, p)
    # every bit of the word, generated (63..32) and hashed (31..0), is set half the time
    n = words.size
    flat = words.reshape(-1)
    for bit in range(64):
        ones = int(((flat >> np.uint64(bit)) & np.uint64(1)).sum())
        assert abs(ones - n / 2) < 5 * np.sqrt(n / 4), (bit, ones, n)


def _pearson_bound(n):
    return 5.0 / np.sqrt(n)


def test_neighbouring_streams_and_slots_are_uncorrelated(words):
    u = _u01(words)
    n = N_STREAMS
    # stream k against stream k + 1 (next bounce), k + 4 (next sample), k + 16 (next pixel), same slot
    for step in (1, 4, 16):
        for slot in (0, 1, 5):
            r = np.corrcoef(u[:-step, slot], u[step:, slot])[0, 1]
            assert abs(r) < _pearson_bound(n - step), (step, slot, r)
    # slots of one stream against each other (same block: 0..3; across blocks: 3 -> 4)
    for a, b in ((0, 1), (1, 2), (2, 3), (3, 4), (0, 4)):
        r = np.corrcoef(u[:, a], u[:, b])[0, 1]
        assert abs(r) < _pearson_bound(n), (a, b, r)


def test_hashed_low_half_is_independent_of_the_generated_high_half(words):
    flat = words.reshape(-1)
    hi = (flat >> np.uint64(32)).astype(np.uint64)
    lo = (flat & np.uint64(0xFFFFFFFF)).astype(np.uint64)
    # joint distribution of the top four bits of each half: 256 cells, chi-square with 255 degrees of freedom
    cell = ((hi >> np.uint64(28)) << np.uint64(4)) | (lo >> np.uint64(28))
    counts = np.bincount(cell.astype(np.int64), minlength=256)
    chi2 = ((counts - flat.size / 256.0) ** 2 / (flat.size / 256.0)).sum()
    assert chi2 < 255 + 5 * np.sqrt(2 * 255), chi2
    # and the low half alone is uniform over 1024 bins
    counts = np.bincount((lo >> np.uint64(22)).astype(np.int64), minlength=1024)
    chi2 = ((counts - flat.size / 1024.0) ** 2 / (flat.size / 1024.0)).sum()
    assert chi2 < 1023 + 5 * np.sqrt(2 * 1023), chi2
    # the low half of the word next to it in the block (the hash mixes a word with its neighbour) does not follow the high half either
    r = np.corrcoef(hi[:-1].astype(np.float64), lo[1:].astype(np.float64))[0, 1]
    assert abs(r) < _pearson_bound(flat.size - 1), r


def test_unit_ball_fields_are_uniform(words):
    """A unit-ball candidate cuts one 64-bit word into three 21-bit coordinates (bits 63..43, 42..22, 21..1): each uniform over 64
    bins, and the three pairwise uncorrelated."""
    flat = words[:, :4].reshape(-1)
    f = [((flat >> np.uint64(s)) & np.uint64(0x1FFFFF)).astype(np.int64) for s in (43, 22, 1)]
    for k in range(3):
        counts = np.bincount(f[k] >> 15, minlength=64)
        chi2 = ((counts - flat.size / 64.0) ** 2 / (flat.size / 64.0)).sum()
        assert chi2 < 63 + 5 * np.sqrt(2 * 63), (k, chi2)
    for a, b in ((0, 1), (1, 2), (0, 2)):
        r = np.corrcoef(f[a].astype(np.float64), f[b].astype(np.float64))[0, 1]
        assert abs(r) < _pearson_bound(flat.size), (a, b, r)
