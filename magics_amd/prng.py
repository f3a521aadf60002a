"""The random stream of the reference's spawner: `GlobalEntropy<bevy_prng::WyRand>` reseeded with
`config.simulation.prng_seed.to_le_bytes()` (crates/magics/src/simulation_loader.rs:651-652) and
sampled through rand's `Rng` methods (spawner.rs:448-449,608-610,631; formation.rs:572;
robot.rs:1599).

Third-party algorithms restated from their published definitions — the crates are absent from
/root/reference, so this stream is UNPINNED by the reference (nothing in its tests fixes a draw):
  * wyrand 0.2.0 (Cargo.lock:8191)  WyRand::rand: state += P0; t = state * (state ^ P1) (128 bit);
    output = hi(t) ^ lo(t); next_u32 = low half; from_seed = u64::from_ne_bytes (little endian here)
  * rand 0.8.5 (Cargo.lock:5732)    UniformFloat<f32>::sample_single (half-open ranges),
    Uniform::new_inclusive(..).sample (inclusive ranges), UniformInt<u32>::sample_single
    (IteratorRandom::choose on an exact-size iterator), Bernoulli (gen_bool), SeedableRng::from_rng
    (ForkableRng::fork_rng draws the 8 seed bytes of the child)
"""
import numpy as np

F = np.float32
_M64 = (1 << 64) - 1
_P0, _P1 = 0xA0761D6478BD642F, 0xE7037ED1A0B428DB


def _f32_from_bits(b):
    return np.array([b], dtype=np.uint32).view(np.float32)[0]


def _bits_from_f32(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


class WyRand:
    def __init__(self, seed):
        self.state = int(seed) & _M64  # from_seed(seed.to_le_bytes()) on a little-endian host

    def next_u64(self):
        self.state = (self.state + _P0) & _M64
        t = self.state * (self.state ^ _P1)
        return ((t >> 64) ^ t) & _M64

    def next_u32(self):
        return self.next_u64() & 0xFFFFFFFF

    # -- rand 0.8.5 ---------------------------------------------------------------------------------
    def _value0_1(self):
        """(rng.gen::<u32>() >> 9).into_float_with_exponent(0) - 1.0: 23 random mantissa bits in [0, 1)."""
        return _f32_from_bits((self.next_u32() >> 9) | (127 << 23)) - F(1.0)

    def gen_range_f32(self, low, high):
        """rng.gen_range(low..high): UniformFloat::<f32>::sample_single."""
        low, high = F(low), F(high)
        if not low < high:
            raise ValueError("UniformSampler::sample_single: low >= high")
        scale = high - low
        while True:
            res = self._value0_1() * scale + low
            if res < high:
                return res
            scale = _f32_from_bits(_bits_from_f32(scale) - 1)  # decrease_masked: the edge case res == high

    def gen_range_f32_inclusive(self, low, high):
        """rng.gen_range(low..=high): Uniform::new_inclusive(low, high).sample(rng)."""
        low, high = F(low), F(high)
        if not low <= high:
            raise ValueError("Uniform::new_inclusive called with `low > high`")
        max_rand = _f32_from_bits((0xFFFFFFFF >> 9) | (127 << 23)) - F(1.0)
        scale = (high - low) / max_rand
        while scale * max_rand + low > high:
            scale = _f32_from_bits(_bits_from_f32(scale) - 1)
        return self._value0_1() * scale + low

    def gen_index(self, ubound):
        """rand::seq::gen_index for ubound <= u32::MAX: UniformInt::<u32>::sample_single(0, ubound)."""
        rng_range = int(ubound) & 0xFFFFFFFF
        if rng_range == 0:
            return self.next_u32()
        lz = 32 - rng_range.bit_length()
        zone = ((rng_range << lz) - 1) & 0xFFFFFFFF
        while True:
            m = self.next_u32() * rng_range
            if (m & 0xFFFFFFFF) <= zone:
                return m >> 32

    def gen_bool(self, p):
        """Bernoulli::new(p).sample: one u64 is drawn unless p == 1."""
        p = float(p)
        if not 0.0 <= p <= 1.0:
            raise ValueError("p is outside range [0.0, 1.0]")
        if p == 1.0:
            return True
        return self.next_u64() < int(p * 18446744073709551616.0)

    def fork(self):
        """ForkableRng::fork_rng: the child's 8 seed bytes are one draw of the parent."""
        return WyRand(self.next_u64())
