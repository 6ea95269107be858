"""How much does the placement of the key buffers matter?  One dummy device allocation of <pad> bytes is made before key generation, then the
bootstrap kernel of each tier is timed (development aid; profiles/r02_exp_ablations.log).  usage: placement_probe.py <pad bytes> [tier names]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
from dctfhe import params as P
from dctfhe.engine import Context, Keys


def main():
    pad = int(sys.argv[1])
    want = sys.argv[2:] or ["T6a", "T4", "T5a"]
    ctx = Context(0)
    hip = C.CDLL("libamdhip64.so")
    ptrs = []
    if pad > 0:
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(pad)) == 0
        ptrs.append(p)
    ps = P.default_params()
    keys = Keys(ctx, P.to_c_params(ps), seed=1)
    out = []
    for i, t in enumerate(ps.tiers):
        if t.name in want:
            out.append(f"{t.name} {keys.bench_pbs(i, 4096 if t.logN < 13 else 2048, reps=2):.2f} ms")
    print(f"pad {pad:>10}: " + ", ".join(out), flush=True)
    keys.close()
    ctx.close()


if __name__ == "__main__":
    main()
