"""Philox4x32-10 in numpy -- the oracle's copy of the engine's counter-based draw spec.

TEST INFRASTRUCTURE ONLY (see oracle/twoarmy_oracle.c header).

Draw spec shared by oracle and engine:  word(seed, env_id, t, slot) = Philox4x32-10(
key=(seed & 0xffffffff, seed >> 32), counter=(env_id, t, slot >> 2, 0x54574F41))[slot & 3],
value = lo + word % n.  `t` counts step() calls of that env since construction.
Slots: 0 gate(10) 1 wall_i1(4) 2 wall_i2(4) 3 spawn(4) 4 coin_a(2) 5 coin_b(2) 8 action(5).
It replaces the reference's single global MT19937 stream (twoarmy_v4.py:117,149,184,190,215,303,310),
whose data-dependent consumption cannot be vectorised; the golden harness replays these words
into the reference by patching np.random.choice (oracle/ref_harness.py).
"""
import numpy as np

DRAW_TAG = 0x54574F41
S_GATE, S_WALL1, S_WALL2, S_SPAWN, S_COIN_A, S_COIN_B, S_ACTION = 0, 1, 2, 3, 4, 5, 8

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(key0, key1, c0, c1, c2, c3):
    """Vectorised over numpy arrays of uint32; returns 4 uint32 arrays."""
    with np.errstate(over="ignore"):
        k0 = np.asarray(key0, dtype=np.uint32)
        k1 = np.asarray(key1, dtype=np.uint32)
        c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = (k0 + _W0).astype(np.uint32)
            k1 = (k1 + _W1).astype(np.uint32)
    return c0, c1, c2, c3


def draw_word(seed, env_id, t, slot):
    out = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, env_id, t, slot >> 2, DRAW_TAG)
    return out[slot & 3]


def draw_words8(seed, env_id, t):
    """All 8 slot words of one (env, t) as a uint32[8] (slots 0..7)."""
    a = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, env_id, t, 0, DRAW_TAG)
    b = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, env_id, t, 1, DRAW_TAG)
    return np.array([int(x) for x in a] + [int(x) for x in b], dtype=np.uint32)


def action_indices(seed, env_ids, t):
    """Policy indices 0..4 for the step-only benchmark stream (slot 8)."""
    return (draw_word(seed, np.asarray(env_ids, dtype=np.uint32), np.asarray(t, dtype=np.uint32), S_ACTION)
            % np.uint32(5)).astype(np.int32)
