"""Noise-model search for the small-key lengths of the exact-evaluation catalogue (dctfhe/params.py::default_params): per group of
tiers that share a key-switch key, the smallest n (steps of 8) that keeps the worst look-up site of the benchmark circuits at the
budget.  The circuits are compiled once (calibration is the slow part); each candidate only re-runs the encoding / tier assignment
and the noise pricing (dctfhe/compile.py::_assign_encodings, _estimate_noise).  CPU only.
usage: python tools/param_search.py [--configs r20_24_16,r20_3_32,r18_3_32]   -> the table profiles/r03_param_search.log holds"""
import argparse
import copy
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
import numpy as np  # noqa: E402

from dctfhe import compile as cc, models, params as P  # noqa: E402

GROUPS = {"table808": ["T6", "T6a", "T5a"], "refresh": ["T4r", "T4r2"], "rescale": ["T4"], "bit": ["B", "Ba", "Ba2"]}


def circuits(names):
    import bench
    out = {}
    for name in names:
        factory, in_ch, img, make_batch, _ = bench.CONFIGS[name]
        t0 = time.time()
        calib = make_batch(16 if name == "r18_48_112" else 100, 7)
        model = getattr(models, factory)(bit_width=4, in_channels=in_ch, img_size=img, seed=0)
        out[name] = cc.compile_model(model, calib, rounding_threshold_bits=6, n_bits=5, p_error=0.01)
        print(f"# compiled {name} in {time.time() - t0:.0f} s: worst site {out[name].worst_site_failure:.2e}, expected failures / image "
              f"{out[name].expected_failures_per_image:.2e}", flush=True)
    return out


def with_n(ps, changes):
    tiers = [dataclasses.replace(t, n=changes.get(t.name, t.n), lwe_sigma=0.0) if t.name in changes else t for t in ps.tiers]
    return dataclasses.replace(ps, tiers=tiers, table_tier_for_w=dict(P.default_params().table_tier_for_w),
                               table_tier_fallback_for_w=dict(P.default_params().table_tier_fallback_for_w))


def price(circs, ps):
    worst, fails, counts = 0.0, {}, {}
    for name, c in circs.items():
        c2 = copy.copy(c)
        c2.tensors = [copy.copy(t) for t in c.tensors]
        c2.ops = [copy.copy(o) for o in c.ops]
        for o in c2.ops:
            o.ip, o.lp = list(o.ip), list(o.lp)
        c2.param_set = copy.deepcopy(ps)
        cc._assign_encodings(c2)
        cc._estimate_noise(c2)
        if getattr(c2.param_set, "table_tier_fallback_for_w", None) and c2.worst_site_failure > c2.param_set.p_budget:
            c2.param_set.table_tier_for_w = {**c2.param_set.table_tier_for_w, **c2.param_set.table_tier_fallback_for_w}
            c2.param_set.table_tier_fallback_for_w = None
            cc._assign_encodings(c2)
            cc._estimate_noise(c2)
        worst = max(worst, c2.worst_site_failure)
        fails[name] = c2.expected_failures_per_image
        counts[name] = c2.pbs_counts()
    return worst, fails, counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="r20_24_16,r20_3_32,r18_3_32")
    ap.add_argument("--budget", type=float, default=1e-12)
    args = ap.parse_args()
    circs = circuits(args.configs.split(","))
    base = P.default_params()
    w0, f0, c0 = price(circs, base)
    print(f"# catalogue as shipped: worst site {w0:.2e}; tiers " + ", ".join(f"{t.name} n={t.n}" for t in base.tiers))
    print("# pbs per image:", {k: v for k, v in c0.items()})
    best = {}
    for gname, names in GROUPS.items():
        n0 = next(t.n for t in base.tiers if t.name == names[0])
        row = []
        for n in range(n0 + 16, n0 - 49, -8):
            w, f, c = price(circs, with_n(base, {nm: n for nm in names}))
            moved = {k: {t: v.get(t, 0) for t in ("B", "Ba", "Ba2", "T4r", "T4r2")} for k, v in c.items()}
            row.append((n, w))
            print(f"{gname:9s} n={n:4d}: worst site {w:.2e}" + ("" if moved == {k: {t: v.get(t, 0) for t in ('B', 'Ba', 'Ba2', 'T4r', 'T4r2')} for k, v in c0.items()} else
                                                                  "   (tier assignment changed: " + str({k: v for k, v in moved.items()}) + ")"), flush=True)
        ok = [n for n, w in row if w <= args.budget]
        best[gname] = min(ok) if ok else n0
    print("# smallest n within the budget, one group at a time:", best)
    allc = {nm: best[g] for g, names in GROUPS.items() for nm in names}
    w, f, c = price(circs, with_n(base, allc))
    print(f"# all groups at once: worst site {w:.2e}, expected failures per image {f}, pbs per image {c}")


if __name__ == "__main__":
    main()
