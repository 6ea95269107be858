"""Per-kernel counter table from rocprofv3 --pmc result databases (one database per pass; ROCm 7.2 writes rocpd sqlite).
usage: pmc_table.py <out.txt> <kernel name fragment> <pass1.db> [pass2.db ...]
Values are summed over all SEs / XCDs and averaged over the launches of the kernel (the first, cold launch included)."""
import collections
import glob
import sqlite3
import sys


def per_kernel(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in con.execute("select * from counters_collection"):
        d = dict(zip(cols, r))
        per[d["kernel_name"].split("(")[0].replace("void ", "").strip()][d["counter_name"]][d["dispatch_id"]] += float(d["value"])
    return per


def main():
    out, frag, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    table = collections.defaultdict(dict)
    for pat in dbs:
        for db in glob.glob(pat):
            for k, counters in per_kernel(db).items():
                if frag in k.replace(" ", ""):
                    for c, disp in counters.items():
                        table[k][c] = (len(disp), sum(disp.values()) / len(disp))
    lines = []
    for k in sorted(table):
        lines.append(k)
        t = table[k]
        for c in sorted(t):
            lines.append(f"  {c:28s} {t[c][1]:.4g}   ({t[c][0]} launches)")
        g = lambda c: t.get(c, (0, 0.0))[1]
        if g("SQ_WAVE_CYCLES"):
            lines.append(f"  VALU-active / wave-cycles = {g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'):.3f} per wave")
            if g("SQ_WAIT_INST_LDS"):
                lines.append(f"  waiting to issue an LDS instruction / wave-cycles = {g('SQ_WAIT_INST_LDS') / g('SQ_WAVE_CYCLES'):.3f}")
            if g("SQ_WAIT_ANY"):
                lines.append(f"  parked at s_waitcnt / s_barrier / wave-cycles = {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f}")
        if g("TCC_HIT_sum") + g("TCC_MISS_sum"):
            lines.append(f"  L2 hit rate = {g('TCC_HIT_sum') / (g('TCC_HIT_sum') + g('TCC_MISS_sum')):.3f}")
        if g("SQ_INSTS_LDS"):
            lines.append(f"  LDS bank-conflict cycles per LDS instruction = {g('SQ_LDS_BANK_CONFLICT') / g('SQ_INSTS_LDS'):.2f}")
        if g("FETCH_SIZE") or g("WRITE_SIZE"):
            lines.append(f"  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB = {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e9:.2f} GB "
                         f"(read {2 * g('FETCH_SIZE') * 1024 / 1e9:.2f}, written {g('WRITE_SIZE') * 1024 / 1e9:.2f})")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
