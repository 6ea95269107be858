"""Parameter tiers and the noise model that justifies them.

The reference hands `p_error` to Concrete's optimizer (homomorphic_eval.py:282, run_homomorphic_eval.sh:26)
and never sees the parameters it picks.  This build owns them: a static catalogue of tiers plus the
textbook TFHE variance formulas (Chillotti et al. 2020, section 4; all variances in torus^2 units),
used by the compiler to price every table site and to refuse a circuit whose failure budget is blown.

Security: binary keys; the minimal noise for ~128-bit security is taken from the linear fit
    log2 sigma_min(d) = -0.02641 * d + 2.49          (floor 2^-62)
through the tfhe-rs 128-bit parameter sets (d = 742, 864, 2048) -- background knowledge, not a
security proof; the catalogue never goes below it.
"""
import math
from dataclasses import dataclass, field, asdict


def sigma_min(d):
    return 2.0 ** max(-62.0, -0.02641 * d + 2.49)


@dataclass
class TierSpec:
    name: str
    n: int
    k: int
    logN: int
    l: int
    beta: int
    lk: int
    betak: int
    ksk_share: int = -1
    unroll: int = 1            # key bits per blind-rotate iteration (2: two-bit rotation, csrc/pbs_core.h; k = 1, l = 1 only)
    key_lds: int = 0           # 1: bootstrap-key tiles shared by the waves of a workgroup through LDS (Ba2's geometry only; no faster: DESIGN 5)
    lwe_sigma: float = 0.0
    glwe_sigma: float = 0.0

    def __post_init__(self):
        if self.lwe_sigma == 0.0:
            self.lwe_sigma = sigma_min(self.n)
        if self.glwe_sigma == 0.0:
            self.glwe_sigma = sigma_min(self.k << self.logN)

    @property
    def N(self):
        return 1 << self.logN

    def as_dict(self):
        d = asdict(self)
        d.pop("name")
        return d


@dataclass
class ParamSet:
    """D: big LWE dimension (master key); tiers by role."""
    D: int
    tiers: list
    bit_tier: int                      # index of the one-bit (rounding) tier
    table_tier_for_w: dict             # table input width w -> tier index (outputs that feed a convolution / pooling)
    coarse_tier_for_w: dict = None     # same, for tables whose output only feeds an add or the circuit output (noisier is fine)
    table_tier_fallback_for_w: dict = None   # quieter (slower) tiers the compiler falls back to when a site leaves the budget
    bit_tier_coarse: int = None        # one-level twin of the bit tier for the last rounding steps of a site
    bit_tier_coarse2: int = None       # a still cheaper (noisier) twin for the steps after those
    refresh_min_w: int = None          # tables this wide that feed a convolution are split: coarse look-up + small-ring refresh
    p_budget: float = 1e-12            # failure probability a single look-up site may spend on cheaper rounding steps
    input_sigma: float = 0.0
    input_dim: int = 0                 # fresh encryptions mask only this prefix of the big key (0 = D); sets input_sigma's dimension
    fft_noise_c: float = 2.0           # empirical constant of the f64-FFT error term (tests/test_gpu_noise.py)

    def __post_init__(self):
        if self.input_sigma == 0.0:
            self.input_sigma = sigma_min(self.input_dim or self.D)

    @property
    def n_max(self):
        return max(t.n for t in self.tiers)

    def tier_for_width(self, w, coarse=False):
        table = self.coarse_tier_for_w if (coarse and self.coarse_tier_for_w) else self.table_tier_for_w
        for ww in sorted(table):
            if w <= ww:
                return table[ww]
        raise ValueError(f"no tier for a table of {w} input bits")


# ------------------------------------------------------------------------------------------ variances
def ks_limbs(t):
    """byte limbs kept per key-switch-key word (csrc/dctfhe.hip ks_limbs): the key lives on the 2^-(8 limbs) torus grid"""
    bits = t.lk * t.betak + 6
    return 2 if bits <= 16 else 4 if bits <= 32 else 8


def var_keyswitch(D_eff, t):
    B = 2.0 ** t.betak
    row = t.lwe_sigma ** 2 + 2.0 ** (-16 * ks_limbs(t)) / 12.0          # the key's body is rounded to its grid: 2^-32 (2^-16 on the one-bit tiers)
    return D_eff * t.lk * ((B * B + 2) / 12.0) * row + (D_eff / 2.0) * 2.0 ** (-2 * t.betak * t.lk) / 12.0


def var_modswitch(t, centred=True):
    """rounding every word of the small ciphertext to 2N levels: sum_i s_i e_i + e_b with e uniform on +-1/(4N).  The engine's mod
    switch is CENTRED (csrc/kernels.h k_ms_center, oracle ref_ms_center): half the sum of the known remainders e_i comes off the body,
    leaving sum_i (s_i - 1/2) e_i -- n/4 units instead of the n/2 of the plain form (binary key, half the bits set)."""
    return ((t.n / 4.0 if centred else t.n / 2.0) + 1.0) / (48.0 * t.N ** 2)


def var_pbs_out(t, fft_c=2.0):
    B = 2.0 ** t.beta
    kN = t.k * t.N
    ext = t.l * (t.k + 1) * t.N * ((B * B + 2) / 12.0) * t.glwe_sigma ** 2 + (kN / 2.0 + 1.0) * 2.0 ** (-2 * t.beta * t.l) / 12.0
    # f64 FFT rounding of the external product, measured on the GPU (tools/noise_probe.py) and on the CPU oracle
    # against the exact schoolbook product (tests/emul): per CMUX  c * (k+1) l N^2 B^2/12 * 2^-106, c ~ 2.
    fft = fft_c * (t.k + 1) * t.l * float(t.N) ** 2 * (B * B / 12.0) * 2.0 ** -106
    if getattr(t, "unroll", 1) == 2:
        # two-bit rotation, per PAIR: three products scaled by (X^e - 1) (variance x2) -> key and FFT noise 6 units instead
        # of 2; the single decomposition error enters through (X^e - 1) b_w, one of the three b_w set with probability
        # 3/4 -> 1.5 units where the one-bit chain has 2 x 1/2 (the formula above counts a unit per CMUX, twice the truth)
        key = t.l * (t.k + 1) * t.N * ((B * B + 2) / 12.0) * t.glwe_sigma ** 2
        dec = (kN / 2.0 + 1.0) * 2.0 ** (-2 * t.beta * t.l) / 12.0
        return (t.n / 2.0) * (6.0 * key + 3.0 * dec + 6.0 * fft)
    return t.n * (ext + fft)


def p_fail(margin, var):
    """two-sided Gaussian tail beyond `margin`"""
    if var <= 0:
        return 0.0
    return math.erfc(margin / math.sqrt(2.0 * var))


# ------------------------------------------------------------------------------------------ catalogue
def default_params():
    """Exact-evaluation set: every table site fails with probability < 1e-12 under the model above.

    Small-key lengths: n = 800 for the 5/6-bit table tiers, 760 / 720 for the 4-bit ones and 560 for the one-bit tiers are the
    smallest (in steps of 8) that keep every site of the benchmark circuits under that budget (tools/param_search.py ->
    profiles/r03_param_search.log; round 3's centred mod switch bought the last 8 bits of each table tier) with base-4 key-switch gadgets
    (below) and a key switch that sums only over the effective dimension of its input (2048 for everything downstream of a
    refresh or of the client, who encrypts under the first 2048 key bits: input_dim); the blind rotation is linear in n.

    T6 (6-bit tables after a rounded accumulator) needs N = 8192: its mod-switch noise must stay 6.4
    sigma inside a 2^-8 half-box.  T5a/T4 serve the 5-bit residual-sum tables and the 4-bit rescale
    tables.  Outputs that feed convolutions (2-norm ~2^6.7) into p-bit accumulators must stay near 2^-23:
    the f64 FFT error (~ N^2 B^2) forces small digits, hence three levels (T6, T4r, T4r2).
    B is the one-bit tier of the rounding chain: margin 1/4, so a small ring, but two levels because its
    output is subtracted from a p-bit accumulator."""
    # Key-switch gadget of the table tiers: base 4, nine levels (betak = 2, lk = 9) instead of base 8, six levels: the key-switch noise
    # lk (B^2 + 2) / 12 sigma^2 drops from 33 to 13.5 units (0.64 bit of sigma), which buys 24 small-key bits per tier -- 808 / 768 / 728
    # instead of 832 / 792 / 752, 3 % of every table bootstrap -- for 1.5x the work of their key switches (0.16 us of 5-25 us per
    # ciphertext).  The one-bit tiers (half-box 1/4) take base 4 at the SAME depth, five levels: only the top 10 bits of a mask word are
    # switched, the truncation (sigma 2^-6.8) stays below the key noise (2^-5.4), the gadget factor drops from 27.5 to 7.5 -- which buys
    # 24 key bits (560 instead of 584) and lets the two-bit-rotation tier Ba2 take every step Ba used to run (ResNet-20, ResNet-18 3x32^2).
    t6 = TierSpec("T6", n=800, k=1, logN=13, l=3, beta=11, lk=9, betak=2)
    # T4 / T4r look up 4-bit values (half-box 2^-6, four times the 6-bit tiers'): they afford a shorter small key -- and the
    # blind rotation is linear in n -- at the price of key-switch keys of their own (prefixes of the same small key, noise of
    # their own dimension).  752 / 792 are the smallest (steps of 8) that keep every site of the benchmark circuits at the
    # worst-site level of the 832-bit tiers (tools: the search behind profiles/r02_param_search.log).
    t4 = TierSpec("T4", n=720, k=1, logN=11, l=1, beta=23, lk=9, betak=2, unroll=2)
    b = TierSpec("B", n=560, k=2, logN=10, l=2, beta=14, lk=5, betak=2)
    # T6a: same ring and input margin as T6, one level: its output (sigma ~2^-13) only ever meets the 2^-7 half-box
    # of the residual-sum table, never a convolution.  Half the transforms of T6 for half of the 6-bit sites.
    t6a = TierSpec("T6a", n=800, k=1, logN=13, l=1, beta=22, lk=9, betak=2, ksk_share=0, unroll=2)
    # Ba: one-level bit tier.  The output of rounding step i is amplified by 2^(p-j) only in the later steps j > i,
    # so the later steps of a chain tolerate sigma ~2^-15; the compiler picks, per
    # site, the first step from which Ba is safe.
    ba = TierSpec("Ba", n=560, k=2, logN=10, l=1, beta=23, lk=5, betak=2, ksk_share=3)
    # T4r: small ring, three levels: turns the noisy 4-bit output of a T6a look-up into a convolution-grade ciphertext
    # (sigma ~2^-25).  T6a + T4r costs ~0.7x of one T6 bootstrap.
    t4r = TierSpec("T4r", n=760, k=1, logN=11, l=3, beta=12, lk=9, betak=2)
    # T4r2: the same refresh with two key bits per iteration (general form of the two-bit rotation, csrc/pbs_core.h): 10 %
    # fewer milliseconds per launch, output 0.7 bit noisier (three external products per pair).  The compiler takes it when every
    # site of the circuit stays inside the budget with it (the ResNet-20 circuits do) and falls back to T4r otherwise (two
    # sites of ResNet-18 3x32^2 would sit at 3.8e-12): ParamSet.table_tier_fallback_for_w.  Shares T4r's key-switch key.
    t4r2 = TierSpec("T4r2", n=760, k=1, logN=11, l=3, beta=12, lk=9, betak=2, ksk_share=1, unroll=2)
    # T5a: one-level twin of the three-level 5-bit tier; the 5-bit residual-sum table is split the same way (T5a + T4r)
    t5a = TierSpec("T5a", n=800, k=1, logN=12, l=1, beta=22, lk=9, betak=2, ksk_share=0, unroll=2)
    # (a table of 5 or 6 input bits that feeds a convolution without the split would run on T6; with refresh_min_w = 5 none does)
    # Ba2: the one-level bit tier on the general two-bit rotation: 49.7 ms per launch of 16 384 against 63.6 (profiles/r02_exp_ablations.log),
    # output 0.5 bit noisier (2^-14.5): the compiler gives it the steps of a chain that can take that (78 % of them), Ba the ones before.
    ba2 = TierSpec("Ba2", n=560, k=2, logN=10, l=1, beta=23, lk=5, betak=2, ksk_share=3, unroll=2)
    return ParamSet(D=8192, tiers=[t6, t4r, t4, b, t6a, ba, t4r2, t5a, ba2], bit_tier=3, table_tier_for_w={4: 6, 6: 0},
                    table_tier_fallback_for_w={4: 1}, coarse_tier_for_w={4: 2, 5: 7, 6: 4}, bit_tier_coarse=5, bit_tier_coarse2=8, refresh_min_w=5,
                    input_dim=2048)


def params_for_p_error(p_error=0.01):
    """Catalogue for tier_policy "p_error" (SURVEY 8f-4): every look-up may fail with probability ~p_error, the regime the
    reference runs in (run_homomorphic_eval.sh:26).  z = 2.6 sigma inside the half-box instead of 7, so a 6-bit table fits
    N = 4096 and a 5-bit one N = 2048; outputs that feed a convolution only have to keep the *next* look-up inside its
    half-box (sigma ~2^-16 after the 2^6.7 amplification), which two levels with 16-bit digits give (sigma ~2^-21) -- no
    refresh bootstraps.  Meant for approximate rounding; with exact rounding the one-bit tiers B / Ba serve as before."""
    if not (1e-6 <= p_error <= 0.05):
        raise ValueError("tier_policy 'p_error' is meant for 1e-6 <= p_error <= 0.05; use the exact catalogue below that")
    f6 = TierSpec("F6", n=864, k=1, logN=12, l=2, beta=16, lk=6, betak=3)
    f5 = TierSpec("F5", n=864, k=1, logN=11, l=2, beta=16, lk=6, betak=3, ksk_share=0)
    t4 = TierSpec("T4", n=864, k=1, logN=11, l=1, beta=23, lk=6, betak=3, ksk_share=0, unroll=2)
    b = TierSpec("B", n=660, k=2, logN=10, l=2, beta=14, lk=5, betak=3)
    t5a = TierSpec("T5a", n=864, k=1, logN=12, l=1, beta=22, lk=6, betak=3, ksk_share=0, unroll=2)
    ba = TierSpec("Ba", n=660, k=2, logN=10, l=1, beta=23, lk=5, betak=3, ksk_share=3)
    return ParamSet(D=8192, tiers=[f6, f5, t4, b, t5a, ba], bit_tier=3, table_tier_for_w={4: 1, 5: 1, 6: 0},
                    coarse_tier_for_w={4: 2, 5: 2, 6: 4}, bit_tier_coarse=5, p_budget=p_error / 4.0, input_dim=2048)


def test_params():
    """Tiny rings for CPU-oracle-sized parity tests (NOT secure: dimensions far below the curve)."""
    t_tab = TierSpec("t", n=48, k=1, logN=10, l=2, beta=12, lk=5, betak=4, lwe_sigma=2.0 ** -26, glwe_sigma=2.0 ** -48)
    t_bit = TierSpec("b", n=40, k=2, logN=8, l=2, beta=10, lk=5, betak=4, lwe_sigma=2.0 ** -24, glwe_sigma=2.0 ** -48)
    return ParamSet(D=1024, tiers=[t_tab, t_bit], bit_tier=1, table_tier_for_w={6: 0}, input_sigma=2.0 ** -55)


def to_c_params(ps):
    from .engine import make_params
    return make_params(ps.D, ps.n_max, [t.as_dict() for t in ps.tiers], ps.input_sigma, ps.input_dim)
