"""CPU-tier checks of the arithmetic the sieve scan (air_rs_amd/csrc/adsb_sieve.inc, A/B build) rests on.

The kernel itself is compared with the oracle on the GPU (tests/ab_cases.py under ADSB_SCAN=sieve); here the claims its header
makes are checked exhaustively in numpy, with the float32 operations the kernel uses:
  * U(n) = RN_f32(2^23 + 9 + 1.125 n) - 2^23 is at least the largest b with floor(sqrt(b)) <= floor(sqrt(n)), for every n the
    i8 path can produce -- so G / L relation bits are a superset of the reference's orderings of truncated roots;
  * the fourteen adjacent-sample taps are implied by the gate (demod.rs:17-57): on synthetic data every offset the oracle's
    gate passes is a sieve candidate;
  * trunc(sqrt_f32(n + 0.5)) = floor(sqrt(n)), and (floor(sqrt(x)))^2 > y  <=>  floor(sqrt(x)) > floor(sqrt(y))."""
import numpy as np

import air_rs_amd as A

G_TAPS = (0, 2, 7, 9, 16, 19, 21, 24)   # m[i+d] >= m[i+d+1]
L_TAPS = (1, 6, 8, 18, 20, 22)          # m[i+d] <= m[i+d+1]
PRE_HI, PRE_LO = (0, 2, 7, 9), (1, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 15)
DF_HI, DF_LO = (16, 19, 21, 23, 24), (17, 18, 20, 22, 25)


def slack_f32(n):
    """2^23 + U(n) exactly as the kernel computes it: one float32 FMA on the bit pattern 0x4B000000 + n."""
    f = (np.asarray(n, dtype=np.int64) + 0x4B000000).astype(np.uint32).view(np.float32)
    # float64 evaluates f * 1.125 - 1048567 exactly (both products fit); one rounding to float32 = the FMA's
    return (f.astype(np.float64) * 1.125 - 1048567.0).astype(np.float32)


def test_slack_is_a_superset_for_every_n():
    n = np.arange(0, 32769, dtype=np.int64)
    u = slack_f32(n).view(np.uint32).astype(np.int64) - 0x4B000000     # integer order of the patterns = numeric order
    m = np.floor(np.sqrt(n.astype(np.float64))).astype(np.int64)
    assert ((m * m <= n) & ((m + 1) * (m + 1) > n)).all()
    top = (m + 1) * (m + 1) - 1                                         # the largest b whose truncated root does not exceed n's
    assert (u >= top).all()
    assert (u >= n).all() and (u <= n + n // 8 + 10).all()              # and it is the tangent it claims to be


def test_exact_root_forms():
    n = np.arange(0, 32769, dtype=np.int64)
    r = np.sqrt((n + 0.5).astype(np.float32)).astype(np.uint32).astype(np.int64)   # v_sqrt_f32 + truncating convert
    m = np.floor(np.sqrt(n.astype(np.float64))).astype(np.int64)
    assert (r == m).all()
    rng = np.random.default_rng(3)
    x, y = rng.integers(0, 32769, 200000), rng.integers(0, 32769, 200000)
    assert (((m[x] * m[x]) > y) == (m[x] > m[y])).all()                 # the PPM bit with one root per pair
    assert ((((m[x] + 1) * (m[x] + 1)) > y) == (m[x] >= m[y])).all()    # the gate with one root per group


def test_gate_implies_all_fourteen_taps():
    cfg = A.synth_default(seed=4242, slot_len=700)
    n_samples = 400_000
    iq = A.synth_fill_host(cfg, A.ADSB_SAMPLE_I8, 0, 0, n_samples).astype(np.int64)
    nn = iq[:, 0] ** 2 + iq[:, 1] ** 2
    m = np.floor(np.sqrt(nn.astype(np.float64))).astype(np.int64)
    u = slack_f32(nn).view(np.uint32).astype(np.int64) - 0x4B000000
    g = u[:-1] >= nn[1:]
    l = u[1:] >= nn[:-1]
    n_off = n_samples - 240
    cand = np.ones(n_off, dtype=bool)
    for d in G_TAPS:
        cand &= g[d:d + n_off]
    for d in L_TAPS:
        cand &= l[d:d + n_off]
    idx = np.arange(n_off)[:, None]
    gate = (m[idx + np.array(PRE_HI)].min(1) >= m[idx + np.array(PRE_LO)].max(1)) & \
           (m[idx + np.array(DF_HI)].min(1) >= m[idx + np.array(DF_LO)].max(1))
    assert gate.sum() > 300                       # the planted frames
    assert not (gate & ~cand).any()               # the sieve loses none of them
    assert cand.sum() < 12 * gate.sum()           # ... and lets a handful per tile through besides


def test_tap_algebra_on_words_equals_the_fourteen_taps():
    """sv_taps (adsb_sieve.inc): Y = L & G>>1, Z = Y & Y>>2, A = G & Y>>1, P = A & Z>>6 & G>>16 & Z>>18 & L>>22 & G>>24 on a 96-bit
    window, restated on Python integers, against the fourteen taps evaluated one by one."""
    rng = np.random.default_rng(11)
    m96 = (1 << 96) - 1
    for _ in range(2000):
        # biased bits: mostly ones, so that all fourteen taps coincide often enough to be seen
        g = sum(int(b) << i for i, b in enumerate(rng.random(96) < 0.9))
        l = sum(int(b) << i for i, b in enumerate(rng.random(96) < 0.9))
        y = l & (g >> 1)
        z = y & (y >> 2)
        a = g & (y >> 1)
        p = a & (z >> 6) & (g >> 16) & (z >> 18) & (l >> 22) & (g >> 24) & m96
        for i in range(64):
            want = all((g >> (i + d)) & 1 for d in G_TAPS) and all((l >> (i + d)) & 1 for d in L_TAPS)
            assert ((p >> i) & 1) == int(want)
