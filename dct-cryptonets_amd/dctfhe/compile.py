"""Circuit compiler: float model + calibration batch -> integer circuit for the HIP engine.

Stands in for what `compile_brevitas_qat_model(model.module.feature, calib_data, rounding_threshold_bits,
n_bits, p_error, ...)` does inside Concrete-ML (reference call site homomorphic_eval.py:276-285): import the
quantised graph, fuse every float sub-graph between two integer linear ops into a per-channel table, give
each accumulator a bit-width from the calibration set, and round accumulators to `rounding_threshold_bits`
before their table (exact method, homomorphic_eval.py:279).  Brevitas/Concrete-ML are not available, so
the quantisers are restated here (per-tensor scales from the calibration batch; weights `bit_width`-bit
narrow range as reference models/backbone.py:217-223; activations as :224-227) and the circuit this
produces is the specification the engine and the oracle are both held to.

Circuit semantics (all integers):
  CONV / ADD / SUMPOOL   exact integer arithmetic on message values
  LUT(p, r, w, signed)   idx = m + (2^(p-1) if signed else 0);  t = (idx + 2^(r-1) * [r>0]) >> r;  y = table[channel][t]
Encodings: a tensor with exponent e holds  phase = value * 2^e  (mod 2^64); a LUT shifts its input up to
e = 63 - p first, so that t sits in the top w+1 bits with the padding bit clear.
"""
import math
import struct
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from . import params as P

OP_CONV, OP_ADD, OP_SUMPOOL, OP_LUT = 1, 2, 3, 4
MAGIC = 0x46544344


# ------------------------------------------------------------------------------------------ quantisers
def weight_quant(w, bits):
    """per-tensor, narrow range (reference backbone.py:217-223: Int8WeightPerTensorFloat, narrow_range=True)"""
    qmax = 2 ** (bits - 1) - 1
    s = float(np.abs(w).max()) / qmax
    if s == 0.0:
        s = 1.0
    return np.clip(np.rint(w / s), -qmax, qmax).astype(np.int64), s


def act_scale(x, signed, bits):
    m = float(np.abs(x).max()) if signed else float(max(x.max(), 0.0))
    if m == 0.0:
        m = 1.0
    return m / ((2 ** (bits - 1) - 1) if signed else (2 ** bits - 1))


def act_quant(x, s, signed, bits):
    lo, hi = (-(2 ** (bits - 1)), 2 ** (bits - 1) - 1) if signed else (0, 2 ** bits - 1)
    return np.clip(np.rint(x / s), lo, hi).astype(np.int64)


def conv_int(q, w, stride, pad):
    """exact integer convolution (values stay far below 2^53, so float64 is exact)"""
    out = F.conv2d(torch.from_numpy(q.astype(np.float64)), torch.from_numpy(w.astype(np.float64)), stride=stride, padding=pad)
    return np.rint(out.numpy()).astype(np.int64)


def _bn_apply(bn, x):
    sh = (1, -1, 1, 1)
    return bn.gamma.reshape(sh) * (x - bn.mean.reshape(sh)) / np.sqrt(bn.var.reshape(sh) + bn.eps) + bn.beta.reshape(sh)


def _bn_calibrate(bn, x):
    if bn.mean is None:
        bn.mean = x.mean(axis=(0, 2, 3))
        bn.var = x.var(axis=(0, 2, 3))


# ------------------------------------------------------------------------------------------ circuit objects
@dataclass
class TensorInfo:
    C: int
    H: int
    W: int
    e: int = None            # encoding exponent
    lo: int = 0
    hi: int = 0              # guaranteed (tables) or calibrated (accumulators) value range
    var: float = 0.0         # noise variance estimate (torus^2)
    deff: int = 0            # mask words beyond this index are zero (nested keys: a bootstrap output of ring k*N has a zero tail)


@dataclass
class OpInfo:
    type: int
    src0: int
    src1: int
    dst: int
    ip: list = field(default_factory=lambda: [0] * 12)
    lp: list = field(default_factory=lambda: [0, 0])
    payload: np.ndarray = None
    # LUT metadata
    p: int = 0
    r: int = 0
    w: int = 0
    signed: bool = False
    table_values: np.ndarray = None   # [ntab, 2^w] integer outputs (before encoding)
    nu2: float = 1.0
    note: str = ""
    pfail: float = 0.0
    coarse_from: int = -1             # rounding steps i >= coarse_from run on the one-level bit tier
    coarse2_from: int = -1            # ... and steps i >= coarse2_from on the two-bit-rotation bit tier (ip[11] = tier << 8 | from)
    sim_sigma: float = 0.0            # modelled noise std at the input of the site's table bootstrap (fraction of the torus)


@dataclass
class CompiledCircuit:
    tensors: list
    ops: list
    input_tensor: int
    output_tensor: int
    in_scale: float
    in_bits: int
    out_scale: float
    out_bits: int
    max_bit_width: int
    param_set: object
    rounding_threshold_bits: int
    n_bits: int
    blob: bytes = b""
    expected_failures_per_image: float = 0.0
    rounding_method: str = "exact"
    expected_boundary_flips_per_image: float = 0.0    # approximate rounding only

    @property
    def e_in(self):
        return self.tensors[self.input_tensor].e

    @property
    def e_out(self):
        return self.tensors[self.output_tensor].e

    def n_in(self):
        t = self.tensors[self.input_tensor]
        return t.C * t.H * t.W

    def n_out(self):
        t = self.tensors[self.output_tensor]
        return t.C * t.H * t.W

    @property
    def worst_site_failure(self):
        """largest modelled failure probability per element over the look-up sites"""
        return max((o.pfail for o in self.ops if o.type == OP_LUT), default=0.0)

    def simulation_sigmas(self):
        """per op, the noise std `simulate` injects at the look-up (0 for the levelled ops)"""
        return [o.sim_sigma if o.type == OP_LUT else 0.0 for o in self.ops]

    def pbs_counts(self):
        """{tier name: programmable bootstraps per image} -- table lookups on the site's table tier, rounding steps
        on the bit tier (steps below coarse_from) or the one-level bit tier (steps from coarse_from on)."""
        ps, out = self.param_set, {}
        for o in self.ops:
            if o.type != OP_LUT:
                continue
            s = self.tensors[o.src0]
            n = s.C * s.H * s.W
            out[ps.tiers[o.ip[4]].name] = out.get(ps.tiers[o.ip[4]].name, 0) + n
            if o.r and not o.ip[9]:
                for st in range(o.r):
                    nm = ps.tiers[step_tier(o, st)].name
                    out[nm] = out.get(nm, 0) + n
        return out

    def report(self):
        """Text dump standing in for `fhe_circuit.mlir` (reference homomorphic_eval.py:309-311)."""
        names = {OP_CONV: "conv2d", OP_ADD: "add", OP_SUMPOOL: "sum_pool", OP_LUT: "round_lut"}
        ps = self.param_set
        lines = [f"// dctfhe circuit: {len(self.ops)} ops, max accumulator bit-width {self.max_bit_width}, D={ps.D}"]
        for i, t in enumerate(ps.tiers):
            lines.append(f"// tier {i} {t.name}: n={t.n} k={t.k} N={t.N} l={t.l} beta={t.beta} lk={t.lk} betak={t.betak} "
                         f"sigma_lwe=2^{math.log2(t.lwe_sigma):.1f} sigma_glwe=2^{math.log2(t.glwe_sigma):.1f}")
        for i, o in enumerate(self.ops):
            s, d = self.tensors[o.src0], self.tensors[o.dst]
            head = f"%{o.dst} = {names[o.type]}(%{o.src0}" + (f", %{o.src1}" if o.type == OP_ADD else "") + ")"
            if o.type == OP_CONV:
                head += f" {{cout={o.ip[0]}, k={o.ip[1]}x{o.ip[2]}, stride={o.ip[3]}, pad={o.ip[4]}, nu2={o.nu2:.0f}}}"
            elif o.type == OP_SUMPOOL:
                head += f" {{k={o.ip[0]}}}"
            elif o.type == OP_LUT:
                head += (f" {{p={o.p}, lsbs_removed={o.r}, table_bits={o.w}, signed={int(o.signed)}, shift={o.ip[3]}, "
                         f"tier={ps.tiers[o.ip[4]].name}" + (", rounding=approximate" if o.ip[9] else "") +
                         (f", bit_tier={ps.tiers[o.ip[5]].name}" if (o.r and not o.ip[9]) else "") +
                         (f", steps>={o.ip[8]}:{ps.tiers[o.ip[7]].name}" if (o.r and not o.ip[9] and o.ip[7] >= 0 and o.ip[8] < o.r) else "") +
                         (f", steps>={o.ip[11] & 255}:{ps.tiers[o.ip[11] >> 8].name}" if (o.r and not o.ip[9] and o.ip[11] >= 0 and (o.ip[11] & 255) < o.r) else "") +
                         f", tables={o.ip[6]}, p_fail/elt={o.pfail:.1e}}}  // {o.note}")
            lines.append(f"{head} : [{s.C}x{s.H}x{s.W}] -> [{d.C}x{d.H}x{d.W}] e={d.e}")
        lines.append(f"// expected table failures per image (noise model): {self.expected_failures_per_image:.2e}")
        if self.rounding_method == "approximate":
            lines.append(f"// approximate rounding: expected boundary flips per image: {self.expected_boundary_flips_per_image:.2e}")
        return "\n".join(lines)


def step_tier(o, i):
    """tier index of rounding step i of look-up op `o`: bit tier, from ip[8] on the coarse twin ip[7], from ip[11] & 255 on ip[11] >> 8"""
    t = o.ip[5]
    if o.ip[7] >= 0 and i >= o.ip[8]:
        t = o.ip[7]
    if o.ip[11] >= 0 and i >= (o.ip[11] & 255):
        t = o.ip[11] >> 8
    return t


class _Act:
    """integer activation during compilation: calibration values, scale, circuit tensor id, guaranteed range"""

    def __init__(self, q, scale, tid, lo, hi):
        self.q, self.scale, self.tid, self.lo, self.hi = q, scale, tid, lo, hi


def _acc_precision(lo, hi, rtb, margin):
    """smallest (p, r, signed) whose padded range holds [lo, hi] (widened by margin) after rounding"""
    lo = int(math.floor(lo * (1 + margin))) if lo < 0 else int(lo)
    hi = int(math.ceil(hi * (1 + margin)))
    signed = lo < 0
    for p in range(1, 40):
        r = max(0, p - rtb)
        half = (1 << (r - 1)) if r > 0 else 0
        if signed:
            ok = -(1 << (p - 1)) <= lo and hi + half <= (1 << (p - 1)) - 1
        else:
            ok = hi + half <= (1 << p) - 1
        if ok:
            return p, r, signed
    raise ValueError("accumulator range too wide")


def lut_index(m, p, r, signed):
    idx = m + ((1 << (p - 1)) if signed else 0)
    if r > 0:
        idx = (idx + (1 << (r - 1))) >> r
    return idx


def lut_centers(p, r, w, signed):
    """accumulator value each table entry stands for"""
    return (np.arange(1 << w, dtype=np.int64) << r) - ((1 << (p - 1)) if signed else 0)


class _Builder:
    def __init__(self, ps, rtb, margin):
        self.ps, self.rtb, self.margin = ps, rtb, margin
        self.tensors, self.ops = [], []
        self.max_bits = 0

    def tensor(self, C, H, W, lo, hi):
        self.tensors.append(TensorInfo(C, H, W, None, lo, hi))
        return len(self.tensors) - 1

    def conv(self, a, layer, bits):
        wq, sw = weight_quant(layer.weight, bits)
        acc = conv_int(a.q, wq, layer.stride, layer.pad)
        Cout, _, KH, KW = wq.shape
        tid = self.tensor(Cout, acc.shape[2], acc.shape[3], int(acc.min()), int(acc.max()))
        op = OpInfo(OP_CONV, a.tid, -1, tid)
        op.ip[:5] = [Cout, KH, KW, layer.stride, layer.pad]
        op.payload = wq.astype(np.int8)
        op.nu2 = float((wq.astype(np.float64) ** 2).sum(axis=(1, 2, 3)).max())
        self.ops.append(op)
        return _Act(acc, a.scale * sw, tid, int(acc.min()), int(acc.max()))

    def add(self, a, b):
        q = a.q + b.q
        tid = self.tensor(*q.shape[1:], a.lo + b.lo, a.hi + b.hi)
        self.ops.append(OpInfo(OP_ADD, a.tid, b.tid, tid))
        return _Act(q, a.scale, tid, a.lo + b.lo, a.hi + b.hi)

    def sum_pool(self, a, K):
        B, C, H, W = a.q.shape
        Ho, Wo = H // K, W // K     # floor mode drops the border (reference backbone.py:276 nn.AvgPool2d)
        q = a.q[:, :, :Ho * K, :Wo * K].reshape(B, C, Ho, K, Wo, K).sum(axis=(3, 5))
        tid = self.tensor(C, Ho, Wo, a.lo * K * K, a.hi * K * K)
        op = OpInfo(OP_SUMPOOL, a.tid, -1, tid)
        op.ip[0] = K
        self.ops.append(op)
        return _Act(q, a.scale, tid, a.lo * K * K, a.hi * K * K)

    def lut_to_conv(self, a, fn, per_channel, rounding, out_scale, note):
        """table site whose output feeds a convolution; wide sites are split into a cheap noisy look-up followed by an
        identity 'refresh' bootstrap on a small ring (ParamSet.refresh_min_w)"""
        y = self.lut(a, fn, per_channel, rounding, out_scale, note)
        op = self.ops[-1]
        if self.ps.refresh_min_w is not None and op.w >= self.ps.refresh_min_w:
            y = self.lut(y, lambda vals: vals, False, False, out_scale, note + " (refresh)")
        return y

    def lut(self, a, fn, per_channel, rounding, out_scale, note):
        """fn(values[ntab or 1, n]) -> integer outputs; values are message values of `a` (ints).
        rounding=True: `a` is an accumulator, calibrated range + rounding to rtb bits;
        rounding=False: `a` has a guaranteed range, table covers it exactly."""
        C = a.q.shape[1]
        if rounding:
            p, r, signed = _acc_precision(int(a.q.min()), int(a.q.max()), self.rtb, self.margin)
        else:
            p, r, signed = _acc_precision(a.lo, a.hi, 64, 0.0)
        w = p - r
        self.max_bits = max(self.max_bits, p)
        centers = lut_centers(p, r, w, signed)
        ntab = C if per_channel else 1
        vals = np.broadcast_to(centers[None, :], (ntab, centers.size))
        table = np.asarray(fn(vals), dtype=np.int64).reshape(ntab, 1 << w)
        idx = lut_index(a.q, p, r, signed)
        if idx.min() < 0 or idx.max() >= (1 << w):
            raise ValueError(f"{note}: calibration values leave the table range")
        ch = np.arange(C).reshape(1, C, 1, 1) if per_channel else np.zeros((1, 1, 1, 1), np.int64)
        q = table[np.broadcast_to(ch, idx.shape), idx]
        lo, hi = int(table.min()), int(table.max())
        tid = self.tensor(*a.q.shape[1:], lo, hi)
        op = OpInfo(OP_LUT, a.tid, -1, tid, p=p, r=r, w=w, signed=signed, table_values=table, note=note)
        self.ops.append(op)
        return _Act(q, out_scale, tid, lo, hi)


def compile_model(model, calib, rounding_threshold_bits=6, n_bits=5, param_set=None, range_margin=0.05, p_error=None,
                  rounding_method="exact", tier_policy="exact"):
    """-> CompiledCircuit.  calib: float [B, C, H, W] calibration inputs (reference: first training batch,
    homomorphic_eval.py:258-261).
    tier_policy "exact" (default): the exact-evaluation catalogue of dctfhe/params.py whatever p_error says (the
    reference hands p_error = 0.01 to Concrete's optimiser, homomorphic_eval.py:282; here outputs then equal the integer
    circuit and the modelled failure estimate is reported).  tier_policy "p_error": the cheaper catalogue whose look-ups
    fail with probability <= p_error each -- stochastic outputs, like the reference's.
    rounding_method "approximate" (README.md:95-114 of the reference, rounding_threshold_bits={"n_bits":..,"method":
    "approximate"}): no one-bit rounding steps, the table bootstrap rounds; inputs next to a rounding boundary may land on
    the neighbouring table entry."""
    if rounding_method not in ("exact", "approximate"):
        raise ValueError(f"rounding_method {rounding_method!r}")
    if tier_policy not in ("exact", "p_error"):
        raise ValueError(f"tier_policy {tier_policy!r}")
    own_catalogue = param_set is None
    if param_set is None:
        param_set = P.params_for_p_error(p_error if p_error is not None else 0.01) if tier_policy == "p_error" else P.default_params()
    ps = param_set
    calib = np.asarray(calib, dtype=np.float64)
    bits = model.bit_width
    bld = _Builder(ps, rounding_threshold_bits, range_margin)
    sgn_lo, sgn_hi = -(2 ** (bits - 1)), 2 ** (bits - 1) - 1
    uq_hi = 2 ** bits - 1
    learned = getattr(model, "act_scales", None) or {}

    def scale_of(key, x, signed):
        """the quantiser's learned scale when the checkpoint carried one (dctfhe.checkpoint), else from the calibration batch"""
        s = learned.get(key)
        return float(s) if s else act_scale(x, signed, bits)

    # quant_inp (client side, in the clear; reference backbone.py:231,241)
    s_in = scale_of("quant_inp", calib, True)
    q0 = act_quant(calib, s_in, True, bits)
    t_in = bld.tensor(*q0.shape[1:], sgn_lo, sgn_hi)
    a = _Act(q0, s_in, t_in, sgn_lo, sgn_hi)

    # stem: conv1 -> bn1 -> [QuantReLU] -> quant_out  (backbone.py:232-261), one fused per-channel table
    acc = bld.conv(a, model.conv1, bits)
    real = acc.q * acc.scale
    _bn_calibrate(model.bn1, real)
    h = _bn_apply(model.bn1, real)
    if model.relu1:
        s_r = scale_of("stem_relu", np.maximum(h, 0), False)
        hq = act_quant(np.maximum(h, 0), s_r, False, bits) * s_r
    else:
        s_r, hq = None, h
    s_q0 = scale_of("stem_quant_out", hq, True)

    def chan_fn(bn, s_acc, post):
        def fn(vals):
            x = _bn_apply(bn, (vals * s_acc)[None, :, :, None])   # vals [C, n] -> [1, C, n, 1]
            return post(x)[0, :, :, 0]
        return fn

    def stem_post(x):
        if s_r is not None:
            x = act_quant(np.maximum(x, 0), s_r, False, bits) * s_r
        return act_quant(x, s_q0, True, bits)

    a = bld.lut_to_conv(acc, chan_fn(model.bn1, acc.scale, stem_post), True, True, s_q0, "stem: bn1+relu+quant_out")

    for bi, blk in enumerate(model.blocks):
        # C1 -> BN1 -> relu1 (u4)                                            backbone.py:94-96
        acc1 = bld.conv(a, blk.C1, bits)
        real1 = acc1.q * acc1.scale
        _bn_calibrate(blk.BN1, real1)
        h1 = np.maximum(_bn_apply(blk.BN1, real1), 0)
        s_r1 = scale_of(("block", bi, "relu1"), h1, False)
        r1 = bld.lut_to_conv(acc1, chan_fn(blk.BN1, acc1.scale, lambda x, s=s_r1: act_quant(np.maximum(x, 0), s, False, bits)), True, True, s_r1,
                     f"block{bi}: BN1+relu1")
        # C2 -> BN2 -> quant_out (s4)                                        backbone.py:97-99
        acc2 = bld.conv(r1, blk.C2, bits)
        real2 = acc2.q * acc2.scale
        _bn_calibrate(blk.BN2, real2)
        g2 = _bn_apply(blk.BN2, real2)
        s_qo = scale_of(("block", bi, "quant_out"), g2, True)
        main_hi, main_lo = sgn_hi * s_qo, sgn_lo * s_qo
        # shortcut                                                           backbone.py:100
        if blk.shortcut is None:
            sc_lo, sc_hi = a.lo * a.scale, a.hi * a.scale
            accs = None
        else:
            accs = bld.conv(a, blk.shortcut, bits)
            reals = accs.q * accs.scale
            _bn_calibrate(blk.BNshortcut, reals)
            gs = _bn_apply(blk.BNshortcut, reals)
            s_qs = scale_of(("block", bi, "BNquant_out"), gs, True)
            sc_lo, sc_hi = sgn_lo * s_qs, sgn_hi * s_qs
        # common integer scale of the residual sum: n_bits signed, guaranteed by construction
        zmax, zmin = 2 ** (n_bits - 1) - 1, -(2 ** (n_bits - 1))
        s_c = max(main_hi + sc_hi, -(main_lo + sc_lo)) / zmax
        while (np.rint(main_hi / s_c) + np.rint(sc_hi / s_c) > zmax) or (np.rint(main_lo / s_c) + np.rint(sc_lo / s_c) < zmin):
            s_c *= 1.01
        u = bld.lut(acc2, chan_fn(blk.BN2, acc2.scale, lambda x, s=s_qo, c=s_c: np.rint(act_quant(x, s, True, bits) * s / c).astype(np.int64)),
                    True, True, s_c, f"block{bi}: BN2+quant_out+rescale")
        if accs is None:
            v = bld.lut(a, lambda vals, s=a.scale, c=s_c: np.rint(vals * s / c).astype(np.int64), False, False, s_c, f"block{bi}: rescale shortcut")
        else:
            v = bld.lut(accs, chan_fn(blk.BNshortcut, accs.scale, lambda x, s=s_qs, c=s_c: np.rint(act_quant(x, s, True, bits) * s / c).astype(np.int64)),
                        True, True, s_c, f"block{bi}: BNshortcut+BNquant_out+rescale")
        z = bld.add(u, v)                                                    # backbone.py:102
        zr = np.maximum(z.q * s_c, 0)
        s_r2 = scale_of(("block", bi, "relu2"), zr, False)
        a = bld.lut_to_conv(z, lambda vals, c=s_c, s=s_r2: act_quant(np.maximum(vals * c, 0), s, False, bits), False, False, s_r2, f"block{bi}: relu2")

    # AvgPool2d(k) as a window sum, then QuantIdentity (s4)                  backbone.py:276-278
    K = model.avgpool_kernel
    pooled = bld.sum_pool(a, K)
    realp = pooled.q * pooled.scale / (K * K)
    s_f = scale_of("final", realp, True)
    out = bld.lut(pooled, lambda vals, s=pooled.scale / (K * K), f=s_f: act_quant(vals * s, f, True, bits), False, True, s_f, "avgpool+QuantIdentity")

    circ = CompiledCircuit(tensors=bld.tensors, ops=bld.ops, input_tensor=t_in, output_tensor=out.tid, in_scale=s_in, in_bits=bits,
                           out_scale=s_f, out_bits=bits, max_bit_width=bld.max_bits, param_set=ps,
                           rounding_threshold_bits=rounding_threshold_bits, n_bits=n_bits, rounding_method=rounding_method)
    _assign_encodings(circ)
    _estimate_noise(circ)
    if getattr(ps, "table_tier_fallback_for_w", None) and circ.worst_site_failure > ps.p_budget:
        # a faster, noisier tier (two-bit refresh) took a site out of the budget: take the quiet twins and price again
        ps.table_tier_for_w = {**ps.table_tier_for_w, **ps.table_tier_fallback_for_w}
        ps.table_tier_fallback_for_w = None
        _assign_encodings(circ)
        _estimate_noise(circ)
    if own_catalogue and tier_policy == "exact" and circ.worst_site_failure > 1e-10:
        import warnings
        warnings.warn(f"dctfhe: a look-up site exceeds the exact-evaluation budget (p_fail/element {circ.worst_site_failure:.1e}); "
                      "outputs may differ from the integer circuit -- see CompiledCircuit.report()")
    circ.blob = _serialize(circ)
    return circ


# ------------------------------------------------------------------------------------------ encodings
def _assign_encodings(circ):
    T, ops = circ.tensors, circ.ops
    req = [None] * len(T)
    req[circ.output_tensor] = 63 - (circ.out_bits + 1)          # signed out_bits value + padding
    for o in reversed(ops):
        if o.type == OP_LUT:
            need = 63 - o.p
        else:
            need = req[o.dst]
        for s in ([o.src0, o.src1] if o.type == OP_ADD else [o.src0]):
            req[s] = need if req[s] is None else min(req[s], need)
    T[circ.input_tensor].e = req[circ.input_tensor]
    T[circ.input_tensor].deff = circ.param_set.input_dim or circ.param_set.D
    # a table whose output is only ever added (or decrypted) tolerates a noisier, cheaper tier
    amplified = [False] * len(T)
    for o in ops:
        if o.type in (OP_CONV, OP_SUMPOOL):
            amplified[o.src0] = True
    for o in reversed(ops):          # an add passes the requirement of its result on to its operands
        if o.type == OP_ADD and amplified[o.dst]:
            amplified[o.src0] = amplified[o.src1] = True
    for o in ops:
        if o.type == OP_LUT:
            T[o.dst].e = req[o.dst]
            shift = (63 - o.p) - T[o.src0].e
            assert shift >= 0
            ps = circ.param_set
            tier = ps.tier_for_width(o.w, coarse=not amplified[o.dst])
            if o.w > ps.tiers[tier].logN - 1:
                raise ValueError("table wider than the ring")
            o.ip[:7] = [o.p, o.r, o.w, shift, tier, ps.bit_tier if o.r > 0 else -1, o.table_values.shape[0]]
            o.ip[7], o.ip[8] = (ps.bit_tier_coarse if ps.bit_tier_coarse is not None else -1), o.r      # refined by _estimate_noise
            o.ip[11] = -1
            o.ip[9] = 1 if (circ.rounding_method == "approximate" and o.r > 0) else 0
            o.ip[10] = T[o.src0].deff
            T[o.dst].deff = ps.tiers[tier].k << ps.tiers[tier].logN
            o.lp[0] = (1 << 62) if o.signed else 0
            enc = (o.table_values.astype(object) * (1 << T[o.dst].e)) % (1 << 64)
            o.payload = np.array(enc, dtype=np.uint64).view(np.int64)
        else:
            T[o.dst].e = T[o.src0].e
            T[o.dst].deff = max(T[o.src0].deff, T[o.src1].deff) if o.type == OP_ADD else T[o.src0].deff
            o.ip[10] = T[o.dst].deff
            if o.type == OP_ADD:
                assert T[o.src1].e == T[o.src0].e, "residual operands must share an encoding"


# ------------------------------------------------------------------------------------------ noise budget
def _estimate_noise(circ):
    ps, T = circ.param_set, circ.tensors
    T[circ.input_tensor].var = ps.input_sigma ** 2
    total, flips = 0.0, 0.0
    for o in circ.ops:
        s = T[o.src0]
        n_elt = s.C * s.H * s.W
        if o.type == OP_CONV:
            T[o.dst].var = o.nu2 * s.var
        elif o.type == OP_ADD:
            T[o.dst].var = s.var + T[o.src1].var
        elif o.type == OP_SUMPOOL:
            T[o.dst].var = o.ip[0] ** 2 * s.var
        else:
            tt = ps.tiers[o.ip[4]]
            v_in0 = s.var * 4.0 ** o.ip[3]
            d_in = s.deff or ps.D                                     # the key switch only sums over the non-zero mask words
            v_tab_in = P.var_keyswitch(d_in, tt) + P.var_modswitch(tt)

            approx = bool(o.ip[9])

            c2 = getattr(ps, "bit_tier_coarse2", None)

            def site_pfail(coarse_from, coarse2_from=None):
                pf_, v_ = 0.0, v_in0
                if approx:
                    # no rounding steps: the low r bits ride along; a failure is noise beyond the half-box.  (The two inputs
                    # next to a rounding boundary, 2 of 2^r, sit half an input unit from it and take the neighbouring
                    # entry far more often: the method's own inexactness, reported apart as boundary flips.)
                    return P.p_fail(2.0 ** -(o.w + 2), v_ + v_tab_in)
                if o.r > 0:
                    for i in range(o.r):
                        step = ps.tiers[o.ip[5]]
                        if i >= coarse_from and o.ip[7] >= 0:
                            step = ps.tiers[o.ip[7]]
                        if coarse2_from is not None and i >= coarse2_from:
                            step = ps.tiers[c2]
                        # the tier that runs the step key-switches to its own small key (own length, own noise) and mod-switches on its ring
                        v_bit_in = P.var_keyswitch(max(d_in, step.k << step.logN), step) + P.var_modswitch(step)
                        pf_ += P.p_fail(0.25, 4.0 ** (o.p - i) * v_ + v_bit_in)
                        v_ += P.var_pbs_out(step, ps.fft_noise_c)
                return pf_ + P.p_fail(2.0 ** -(o.w + 2), v_ + v_tab_in)

            pf = site_pfail(o.r)
            if o.r > 0 and o.ip[7] >= 0 and not approx:
                # earliest step from which the one-level bit tier keeps the site within 2x of its all-precise failure rate
                budget = max(2.0 * pf, getattr(ps, "p_budget", 1e-12))
                cf = o.r
                while cf > 0 and site_pfail(cf - 1) <= budget:
                    cf -= 1
                o.coarse_from = o.ip[8] = cf
                pf = site_pfail(cf)
                if c2 is not None:      # ... and, inside that budget, the earliest step from which the two-bit-rotation tier will do
                    cf2 = o.r
                    while cf2 > cf and site_pfail(cf, cf2 - 1) <= budget:
                        cf2 -= 1
                    o.coarse2_from = cf2
                    o.ip[11] = (c2 << 8) | cf2 if cf2 < o.r else -1
                    pf = site_pfail(cf, cf2 if cf2 < o.r else None)
            o.pfail = pf
            v_sim = v_in0
            if o.r > 0 and not approx:                      # what the rounding steps leave on the working ciphertext
                bt = ps.tiers[o.ip[5]]
                for i in range(o.r):
                    v_sim += P.var_pbs_out(ps.tiers[step_tier(o, i)], ps.fft_noise_c)
            o.sim_sigma = math.sqrt(v_sim + v_tab_in)
            if approx:
                flips += (2.0 / 2 ** o.r) * P.p_fail(2.0 ** -(o.p + 2), v_in0 + v_tab_in) * n_elt
            total += pf * n_elt
            T[o.dst].var = P.var_pbs_out(tt, ps.fft_noise_c)
    circ.expected_failures_per_image = total
    circ.expected_boundary_flips_per_image = flips


# ------------------------------------------------------------------------------------------ blob
def _serialize(circ):
    nT, nO = len(circ.tensors), len(circ.ops)
    head = struct.pack("<IIiiiiii", MAGIC, 1, nT, nO, circ.input_tensor, circ.output_tensor, circ.max_bit_width, 0)
    tens = b"".join(struct.pack("<iiii", t.C, t.H, t.W, 0) for t in circ.tensors)
    off = len(head) + len(tens) + nO * 96
    payloads, recs = [], []
    for o in circ.ops:
        pl = b"" if o.payload is None else np.ascontiguousarray(o.payload).tobytes()
        pad = (-len(pl)) % 16
        recs.append(struct.pack("<iiii12i2qqq", o.type, o.src0, max(o.src1, 0), o.dst, *[int(x) for x in o.ip],
                                *[int(x) - (1 << 64) if int(x) >= (1 << 63) else int(x) for x in o.lp], off if pl else 0, len(pl)))
        payloads.append(pl + b"\0" * pad)
        off += len(pl) + pad
    return head + tens + b"".join(recs) + b"".join(payloads)
