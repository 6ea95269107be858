"""The operator surface the reference drives, re-hosted on the HIP engine.

    reference call site                                         here
    compile_brevitas_qat_model(...)  homomorphic_eval.py:276    compile_brevitas_qat_model -> QuantizedModule
    compile_torch_model(...)         homomorphic_eval.py:287    compile_torch_model (same circuit builder)
    Configuration(...)               homomorphic_eval.py:266    Configuration (progress flags kept, inert)
    q.fhe_circuit.graph.maximum_integer_bit_width()    :301     FHECircuit.graph.maximum_integer_bit_width()
    q.fhe_circuit.mlir                                 :311     FHECircuit.mlir  (text dump of the compiled circuit)
    q.fhe_circuit.keygen()                             :315     FHECircuit.keygen()  (keys generated on the GPU)
    q.forward(x, fhe="simulate"|"execute")             :70      QuantizedModule.forward

fhe="execute": quantise -> encrypt -> circuit on ciphertexts -> decrypt -> dequantise, all ciphertext work in
libdctfhe.so.  fhe="simulate" / "disable": the same integer circuit on plaintext phases (1-word
"ciphertexts"), also on the GPU through the same scheduler -- the noise-free circuit.  There is no CPU path.
"""
import time

import numpy as np

from . import compile as cc
from . import params as P
from .engine import Circuit, Context, Keys, Session


class Configuration:
    """Stand-in for concrete.fhe.Configuration (reference homomorphic_eval.py:266-273)."""

    def __init__(self, show_progress=False, progress_tag=False, progress_title="", **kwargs):
        self.show_progress, self.progress_tag, self.progress_title = show_progress, progress_tag, progress_title
        self.extra = kwargs


class _Graph:
    def __init__(self, circ):
        self._c = circ

    def maximum_integer_bit_width(self):
        return self._c.max_bit_width


class FHECircuit:
    def __init__(self, owner):
        self._o = owner
        self.graph = _Graph(owner.compiled)

    @property
    def mlir(self):
        return self._o.compiled.report()

    def keygen(self, seed=None, force=False):
        """reference homomorphic_eval.py:315.  seed: None = 32 fresh bytes from the OS (the default); 32 bytes = a persisted /
        broadcast key seed; int = deterministic test seed (dctfhe.engine.seed_bytes)."""
        self._o._keygen(seed, force)

    # -- client / server split: the owner's methods (QuantizedModule) -----------------------------------------------
    def export_evaluation_keys(self):
        return self._o.export_evaluation_keys()

    def load_evaluation_keys(self, blob):
        return self._o.load_evaluation_keys(blob)

    def evaluate_encrypted(self, cts, batch, dim=None):
        return self._o.evaluate_encrypted(cts, batch, dim)

    @property
    def statistics(self):
        """Concrete's `fhe_circuit.statistics` is a property; here the engine's dctfhe_stats of the compiled circuit"""
        return self._o.statistics()


class QuantizedModule:
    def __init__(self, compiled, device=0, verbose=False, classifier=None):
        self.compiled = compiled
        self.device = device
        self.verbose = verbose
        self._ctx = None
        self._circuit = None
        self._keys = None
        self._sessions = {}
        self.fhe_circuit = FHECircuit(self)
        self.last_timing = None
        self.last_io = None
        self.sim_seed = 977

    # -- lazy device objects -------------------------------------------------------------
    def _context(self):
        if self._ctx is None:
            self._ctx = Context(self.device)
            self._circuit = Circuit(self._ctx, self.compiled.blob)
        return self._ctx

    def _keygen(self, seed, force=False):
        ctx = self._context()
        if self._keys is not None and not force:
            return
        if self._keys is not None:
            for k in [k for k in self._sessions if k[0] == "execute"]:
                self._sessions.pop(k).close()
            self._keys.close()
        self._keys = Keys(ctx, P.to_c_params(self.compiled.param_set), seed)

    def _session(self, mode, batch):
        key = (mode, batch)
        if key not in self._sessions:
            ctx = self._context()
            if mode == "execute" and self._keys is None:
                self._keygen(None)
            self._sessions[key] = Session(ctx, self._circuit, self._keys if mode == "execute" else None, batch)
        return self._sessions[key]

    # -- client / server split (reference homomorphic_eval.py:313-317 keeps both halves in one process) ------------
    def export_evaluation_keys(self):
        """client side: the evaluation keys as a flat uint8 blob to ship to the server (no secret inside)"""
        if self._keys is None:
            self._keygen(None)
        return self._keys.eval.to_blob()

    def load_evaluation_keys(self, blob):
        """server side: evaluate with keys a client generated elsewhere; this module can then run `evaluate_encrypted`
        but can neither encrypt nor decrypt"""
        from .engine import EvalKeys
        ctx = self._context()
        for k in [k for k in self._sessions if k[0] == "execute"]:
            self._sessions.pop(k).close()
        if self._keys is not None:
            self._keys.close()
        self._keys = EvalKeys.from_blob(ctx, blob)

    def evaluate_encrypted(self, cts, batch, dim=None):
        """server side: input ciphertexts [batch * n_in, D+1] -> output ciphertexts [batch * n_out, D+1]; dim: the compact wire
        form instead -- input rows of dim mask words + body, output rows of Session.dims()[1] mask words + body"""
        sess = self._session("execute", batch)
        sess.upload(cts, dim)
        sess.run()
        if dim is None:
            return sess.download().reshape(-1, self._keys.D + 1)
        out_dim = sess.dims()[1]
        return sess.download(out_dim).reshape(-1, out_dim + 1)

    def statistics(self):
        self._context()
        return self._circuit.stats(P.to_c_params(self.compiled.param_set))

    # -- quantisation at the boundary ----------------------------------------------------
    def quantize_input(self, x):
        c = self.compiled
        return cc.act_quant(np.asarray(x, np.float64), c.in_scale, True, c.in_bits)

    def encode_input(self, q):
        return (q.astype(np.int64).astype(np.uint64) << np.uint64(self.compiled.e_in)).reshape(q.shape[0], -1)

    def decode_output(self, phases):
        e = self.compiled.e_out
        v = (phases + (np.uint64(1) << np.uint64(e - 1))).view(np.int64) >> np.int64(e)     # signed, rounded
        return v

    def dequantize_output(self, q):
        return q.astype(np.float64) * self.compiled.out_scale

    # -- the reference's entry point -----------------------------------------------------
    def forward(self, x, fhe="disable"):
        """x: float [B, C, H, W] -> float [B, F]  (reference homomorphic_eval.py:70)."""
        if fhe not in ("disable", "simulate", "execute"):
            raise ValueError(f"fhe mode {fhe!r}")
        x = np.asarray(x)
        q = self.quantize_input(x)
        out_q = self.forward_quantized(q, fhe)
        return self.dequantize_output(out_q)

    def forward_quantized(self, q, fhe="disable"):
        phases = self.encode_input(q)
        B = q.shape[0]
        mode = "execute" if fhe == "execute" else "clear"
        sess = self._session(mode, B)
        if mode == "clear":
            # "simulate" = the integer circuit with the compiler's noise model sampled at every look-up (reference: Concrete's
            # simulation, homomorphic_eval.py:333-347); "disable" = noise-free.  At the exact tiers the two coincide.
            if fhe == "simulate":
                sess.set_noise(self.sim_seed, self.compiled.simulation_sigmas())
                self.sim_seed += 1
            else:
                sess.set_noise(0, None)
        t0 = time.time()
        if mode == "execute":
            # ciphertexts travel in the compact wire form: a fresh encryption masks input_dim words, an output the ring of the last
            # table tier -- not the D words of the master key (include/dctfhe.h dctfhe_encrypt_rows)
            in_dim, out_dim = sess.dims()
            t1 = time.time()
            cts = self._keys.encrypt(phases.reshape(-1), in_dim)
            t2 = time.time()
            sess.upload(cts, in_dim)
            t3 = time.time()
            timing = sess.run(timing=True)
            t4 = time.time()
            out = sess.download(out_dim).reshape(-1, out_dim + 1)
            t5 = time.time()
            out_ph = self._keys.decrypt(out, out_dim).reshape(B, -1)
            self.last_io = dict(encrypt_s=t2 - t1, upload_s=t3 - t2, run_s=t4 - t3, download_s=t5 - t4, decrypt_s=time.time() - t5,
                                input_bytes=int(cts.nbytes), output_bytes=int(out.nbytes))
        else:
            sess.upload(phases)
            timing = sess.run(timing=True)
            out_ph = sess.download().reshape(B, -1)
        self.last_timing = dict(total_ms=timing.total_ms, pbs_ms=list(timing.pbs_ms), ks_ms=timing.ks_ms, linear_ms=timing.linear_ms,
                                wall_s=time.time() - t0)
        return self.decode_output(out_ph)

    def close(self):
        for s in self._sessions.values():
            s.close()
        self._sessions = {}
        if self._keys is not None:
            self._keys.close()
            self._keys = None
        if self._circuit is not None:
            self._circuit.close()
            self._circuit = None
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


def _as_numpy(t):
    if hasattr(t, "detach"):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def compile_brevitas_qat_model(torch_model, torch_inputset, n_bits=5, configuration=None, rounding_threshold_bits=6, p_error=None,
                               verbose=False, device=0, param_set=None, tier_policy="exact", **kwargs):
    """Same keyword surface as the call at reference homomorphic_eval.py:276-285.  `torch_model` is what the reference
    passes -- the trunk `model.module.feature`, a torch.nn.Module walked by dctfhe.torch_import (duck-typed: the float
    `ResNetDCT` and, where Brevitas exists, `ResNetQDCT` import unchanged) -- or a dctfhe.models.ResNetQ description.
    `bit_width` (dctfhe addition): weight/activation width when the module does not say (`qconv_args`).
    rounding_threshold_bits: int (exact rounding) or {"n_bits": int, "method": "exact"|"approximate"} as the reference's
    README.md:95-114 suggests.  tier_policy: "exact" (default, outputs equal the integer circuit whatever p_error) or
    "p_error" (cheaper tiers whose look-ups fail with probability <= p_error; dctfhe addition)."""
    method = "exact"
    if isinstance(rounding_threshold_bits, dict):
        method = str(rounding_threshold_bits.get("method", "exact")).lower().split(".")[-1]      # also accepts "Exactness.APPROXIMATE"
        rtb = rounding_threshold_bits["n_bits"]
    else:
        rtb = rounding_threshold_bits
    from . import torch_import
    bit_width = kwargs.pop("bit_width", None)
    if torch_import.is_torch_module(torch_model):
        torch_model = torch_import.from_torch_module(torch_model, bit_width=bit_width or 4)
    compiled = cc.compile_model(torch_model, _as_numpy(torch_inputset), rounding_threshold_bits=rtb, n_bits=n_bits,
                                param_set=param_set, p_error=p_error, rounding_method=method, tier_policy=tier_policy)
    return QuantizedModule(compiled, device=device, verbose=verbose)


def compile_torch_model(torch_model, torch_inputset, n_bits=5, configuration=None, rounding_threshold_bits=6, p_error=None,
                        verbose=False, device=0, param_set=None, tier_policy="exact", **kwargs):
    """PTQ twin of the above (reference homomorphic_eval.py:287-295): a float torch trunk (`ResNetDCT`), every weight and
    activation quantised to `n_bits` post-training ([K] Concrete-ML's PTQ applies n_bits to all ops) unless `bit_width`
    says otherwise; the circuit builder is the same."""
    from . import torch_import
    if torch_import.is_torch_module(torch_model):
        kwargs.setdefault("bit_width", n_bits)
    return compile_brevitas_qat_model(torch_model, torch_inputset, n_bits, configuration, rounding_threshold_bits, p_error, verbose,
                                      device, param_set, tier_policy, **kwargs)
