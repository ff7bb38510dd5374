"""PCG32 as the reference defines it (geometry/src/rng.rs:5-67), for host-side
scene generation.  ``Rng`` is the scalar generator; ``streams`` runs one
generator per array element (sequence index = element id) with numpy."""
import numpy as np

_STATE = 0x853C49E6748FEA9B
_MULT = 0x5851F42D4C957F2D
_M64 = (1 << 64) - 1
_ONE_MINUS_EPS = np.float32(1.0) - np.float32(2.0 ** -23)
_STEP = np.float32(2.0 ** -32)


class Rng:
    def __init__(self, sequence_index=0):
        self.state = 0
        self.inc = 1
        self.set_sequence(sequence_index)

    def set_sequence(self, index):  # rng.rs:33-39
        self.state = 0
        self.inc = ((index << 1) | 1) & _M64
        self.uniform_u32()
        self.state = (self.state + _STATE) & _M64
        self.uniform_u32()

    def uniform_u32(self):  # rng.rs:42-48
        old = self.state
        self.state = (old * _MULT + self.inc) & _M64
        x = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((x >> rot) | (x << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform_u32_limit(self, limit):  # rng.rs:51-59
        threshold = ((~limit + 1) & 0xFFFFFFFF) % limit
        while True:
            r = self.uniform_u32()
            if r >= threshold:
                return r % limit

    def uniform_float(self):  # rng.rs:63-67 (product formed in f32)
        num = np.float32(self.uniform_u32()) * _STEP
        return float(min(_ONE_MINUS_EPS, num))


def streams(first_sequence, count, draws):
    """float32 array [count, draws]: row i holds the first ``draws`` values of
    uniform_float() from Rng(first_sequence + i)."""
    with np.errstate(over="ignore"):
        idx = np.arange(count, dtype=np.uint64) + np.uint64(first_sequence)
        inc = (idx << np.uint64(1)) | np.uint64(1)
        mult = np.uint64(_MULT)
        state = np.zeros(count, dtype=np.uint64)
        state = state * mult + inc
        state = state + np.uint64(_STATE)
        state = state * mult + inc
        out = np.empty((count, draws), dtype=np.float32)
        for k in range(draws):
            old = state
            state = old * mult + inc
            x = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
            rot = (old >> np.uint64(59)).astype(np.uint32)
            r = (x >> rot) | (x << ((np.uint32(32) - rot) & np.uint32(31)))
            out[:, k] = np.minimum(_ONE_MINUS_EPS, r.astype(np.float32) * _STEP)
    return out
