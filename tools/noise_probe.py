"""Measures, per full-size tier, the noise the GPU bootstrap leaves on its output and the failure margin
of its input (KS + mod switch), against the model in dctfhe/params.py (development aid / calibration)."""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
import numpy as np
from dctfhe import params as P
from dctfhe.engine import Context, Keys

def cent(x): return x.astype(np.int64).astype(np.float64) / 2.0 ** 64

def main():
    ps = P.default_params()
    ctx = Context(0)
    keys = Keys(ctx, P.to_c_params(ps), seed=1)
    count = 2048
    rng = np.random.default_rng(0)
    for ti, t in enumerate(ps.tiers):
        w = 4
        msgs = rng.integers(0, 16, count).astype(np.uint64)
        cts = keys.encrypt(msgs << np.uint64(59))
        t0 = time.time()
        small = keys.keyswitch(ti, cts)
        tks = time.time() - t0
        table = (np.arange(16, dtype=np.int64)) << 57      # identity, output at 2^57 (6 bits of room)
        t0 = time.time()
        out = keys.pbs(ti, small, table, w)
        tp = time.time() - t0
        ph = keys.decrypt(out)
        err = cent(ph - (msgs << np.uint64(57)))
        bad = int((np.abs(err) > 2.0 ** -8).sum())
        good = np.abs(err) < 2.0 ** -8
        sd = err[good].std()
        model = math.sqrt(P.var_pbs_out(t, ps.fft_noise_c))
        print(f"tier {t.name}: wrong={bad}/{count}  out sigma=2^{math.log2(sd):.2f} (model 2^{math.log2(model):.2f})  max|err|=2^{math.log2(np.abs(err[good]).max()):.2f}  ks {tks:.2f}s pbs {tp:.2f}s", flush=True)
        # second application on top (feeds its own output back): checks out noise does not blow a 6-bit input
    keys.close()

if __name__ == "__main__":
    main()
