"""Pins the oracle's RNG contract to independent sources:
libstdc++'s own minstd_rand/uniform_real_distribution (the objects the
reference holds, mhmcmc.h:27-28,55-56,85), the Random123 Philox known-answer
vectors, and scipy's normal quantile for AS241."""
import os
import subprocess

import numpy as np
import pytest
from scipy import stats

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(os.path.dirname(HERE), "oracle", "rng_probe")


def test_minstd_canonical_matches_libstdcxx(orc):
    for seed in (1, 12345, 2147483646, 987654321):
        out = subprocess.check_output([PROBE, str(seed), "64"]).decode().split("\n")
        ref = [float.fromhex(x) for x in out[:64]]
        got = orc.minstd_canonical_stream(seed, 64)
        assert got == ref          # bit-exact
        assert out[64] == "kat10000 399268537"


def test_minstd_survey_values(orc):
    got = orc.minstd_canonical_stream(12345, 4)
    want = [0.72558467636288126, 0.94121549491786571, 0.72023319492750204, 0.41355216605758932]
    assert got == want


def test_philox_random123_kat(orc):
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xffffffff
    assert orc.philox([f, f, f, f], [f, f]) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                      [0xa4093822, 0x299f31d0]) == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_dlog_close_to_libm(orc):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(2000), 10.0 ** rng.uniform(-300, 300, 2000), [1.0, 2.0, 0.5, 2.0 ** -52]])
    got = np.array([orc.lib().orc_dlog(float(v)) for v in x])
    want = np.log(x)
    assert np.all(np.abs(got - want) <= 4e-16 * np.maximum(1.0, np.abs(want)))


def test_ppnd16_matches_scipy(orc):
    rng = np.random.default_rng(2)
    p = np.concatenate([rng.random(4000), [1e-300, 1e-20, 1e-10, 0.075, 0.5, 0.925, 1 - 1e-10]])
    got = np.array([orc.lib().orc_ppnd16(float(v)) for v in p])
    want = stats.norm.ppf(p)
    assert np.all(np.abs(got - want) <= 1e-14 * np.maximum(1.0, np.abs(want)))


def test_normal_stream_moments(orc):
    z = np.array([orc.normal(7, i, 3, 5, 2) for i in range(20000)])
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1) < 0.03
    assert abs(stats.skew(z)) < 0.06 and abs(stats.kurtosis(z)) < 0.12
    # addressing: a different chain / proposal / tag gives a different stream
    assert orc.normal(7, 0, 3, 5, 2) != orc.normal(7, 0, 4, 5, 2)
    assert orc.normal(7, 0, 3, 5, 2) != orc.normal(7, 0, 3, 6, 2)
    assert orc.normal(7, 0, 3, 5, 2) == orc.normal(7, 0, 3, 5, 2)
