"""Synthetic workload: artest's white-noise generator (reference artest.c:744-754), vectorised.

state <- ((state << 4) - state) ^ 1, three times per sample, sample = int32(state >> 32) / 2^32.
Three steps are an affine map whose sign depends on the parity of the state (which alternates every
step), so two samples are one fixed affine map  s -> A2*s + B2 (mod 2^64); numpy's wrapping uint64
cumprod/cumsum evaluates the closed form.  Bit-identical to the scalar generator (tested)."""
import numpy as np

SEED = 0x3141592653589793
_M = (1 << 64) - 1


def _step3(s):
    for _ in range(3):
        s = (((s << 4) - s) ^ 1) & _M
    return s


def noise(count, state=SEED):
    """returns (float32[count] in [-0.5, 0.5), next_state)"""
    if count == 0:
        return np.zeros(0, np.float32), state
    # derive the two-sample affine map from three probes of the scalar recurrence
    s0 = state
    s1 = _step3(s0)
    s2 = _step3(s1)
    # odd/even sub-sequences: s_{k+2} = A*s_k + B (parity of s_k fixed within a sub-sequence)
    A = pow(3375, 2, 1 << 64)
    B_even = (s2 - A * s0) & _M
    s3 = _step3(s2)
    B_odd = (s3 - A * s1) & _M
    n_pairs = (count + 1) // 2 + 1
    with np.errstate(over="ignore"):
        powers = np.cumprod(np.concatenate((np.ones(1, np.uint64), np.full(n_pairs - 1, A, np.uint64))))      # A^k
        geo = np.concatenate((np.zeros(1, np.uint64), np.cumsum(powers[:-1], dtype=np.uint64)))          # 1+A+..+A^(k-1)
        even = powers * np.uint64(s0) + geo * np.uint64(B_even)      # s0, s2, s4, ...
        odd = powers * np.uint64(s1) + geo * np.uint64(B_odd)        # s1, s3, s5, ...
    seq = np.empty(2 * n_pairs, np.uint64)
    seq[0::2], seq[1::2] = even, odd
    states = seq[1:count + 1]
    vals = (states >> np.uint64(32)).astype(np.uint32).view(np.int32).astype(np.float64) / 4294967296.0
    return vals.astype(np.float32), int(states[-1])
