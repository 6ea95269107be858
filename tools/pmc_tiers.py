"""Launches the bootstrap kernel of each tier of the exact-evaluation catalogue a few times (random inputs: the kernel's work
does not depend on the values) -- the program rocprofv3 profiles for the per-tier counter tables under profiles/.
usage: pmc_tiers.py [tier names ...]      (default: every tier the ResNet-20 circuit uses)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
from dctfhe import params as P
from dctfhe.engine import Context, Keys

# ciphertexts per launch as the ResNet-20 24x16^2 circuit issues them (one activation tensor, or the scheduler's 16384 chunk)
COUNTS = {"T6a": 12288, "T4r": 12288, "T4r2": 12288, "Ba": 16384, "Ba2": 16384, "B": 12288, "T5a": 12288, "T4": 12288}


def main():
    want = sys.argv[1:] or list(COUNTS)
    ps = P.default_params()
    ctx = Context(0)
    keys = Keys(ctx, P.to_c_params(ps), seed=1)
    for i, t in enumerate(ps.tiers):
        if t.name in want:
            ms = keys.bench_pbs(i, COUNTS[t.name], reps=2)
            print(f"{t.name}: pbs_kernel<{t.logN},{t.k},{t.l}> unroll {t.unroll}, {COUNTS[t.name]} cts, {ms:.2f} ms per launch", flush=True)
    keys.close()
    ctx.close()


if __name__ == "__main__":
    main()
