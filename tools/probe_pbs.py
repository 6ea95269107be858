"""Stand-alone timing of the bootstrap kernel per tier on the GPU (development aid)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dct-cryptonets_amd"))
from dctfhe.engine import Context, Keys, make_params

def main():
    ctx = Context(0)
    print("fp64 peak TF:", ctx.fp64_peak(), flush=True)
    sets = {
      "T6 N=8192 l=2": dict(n=864, k=1, logN=13, l=2, beta=17, lk=6, betak=3, lwe_sigma=2.0**-20.3, glwe_sigma=2.0**-62),
      "T5 N=4096 l=2": dict(n=864, k=1, logN=12, l=2, beta=16, lk=6, betak=3, lwe_sigma=2.0**-20.3, glwe_sigma=2.0**-62),
      "T4 N=2048 l=1": dict(n=864, k=1, logN=11, l=1, beta=23, lk=6, betak=3, lwe_sigma=2.0**-20.3, glwe_sigma=2.0**-51.6),
      "B  N=1024 k=2 l=2": dict(n=650, k=2, logN=10, l=2, beta=14, lk=4, betak=3, lwe_sigma=2.0**-14.7, glwe_sigma=2.0**-51.6),
      "B' N=2048 k=1 l=2": dict(n=650, k=1, logN=11, l=2, beta=14, lk=4, betak=3, lwe_sigma=2.0**-14.7, glwe_sigma=2.0**-51.6),
    }
    only = sys.argv[1:] 
    for name, t in sets.items():
        if only and not any(o in name for o in only): continue
        t0 = time.time()
        keys = Keys(ctx, make_params(8192, 864, [t], 2.0**-62), seed=1)
        tk = time.time() - t0
        N = 1 << t["logN"]
        T = N // 2 // 16 if not (t["k"] == 2) else N // 2 // 8
        G = max(1, 256 // T)
        res = {}
        for count in (256 * G, 1024 * G):
            ms = keys.bench_pbs(0, count, reps=2)
            res[count] = ms
        M = N / 2
        import math
        fft = 5 * M * math.log2(M)
        fl = t["n"] * ((t["k"] + 1) * t["l"] * fft + (t["k"] + 1) * fft + (t["k"] + 1) ** 2 * t["l"] * M * 8)
        c = max(res)
        print(f"{name}: keygen {tk:.1f}s; " + ", ".join(f"{k} cts: {v:.1f} ms" for k, v in res.items()) +
              f"; {c / res[c] * 1e3:.0f} PBS/s; {fl * c / res[c] / 1e9:.2f} TFLOP/s", flush=True)
        keys.close()

if __name__ == "__main__":
    main()
