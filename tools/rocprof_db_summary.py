"""Summaries of rocprofv3 result databases (ROCm 7.2 writes rocpd sqlite by default).
usage: rocprof_db_summary.py stats <kernel_trace.db> <out.csv>
       rocprof_db_summary.py hbm <fetch.db> <write.db> <out.json> [note]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB: gfx950 reports half of the bytes of a wide coalesced read stream
(calibrated on k_affine, which reads and writes the same byte count)."""
import collections, csv, json, sqlite3, sys


def stats(db, out):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(top_kernels)")]
    rows = list(con.execute("select * from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(cols)
        w.writerows(rows)
    for r in rows[:10]:
        print([str(x)[:64] for x in r])


def per_kernel(db):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in con.execute("select * from counters_collection"):
        d = dict(zip(cols, r))
        per[d["kernel_name"].split("(")[0].replace("void ", "").strip()][d["dispatch_id"]] += float(d["value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}


def hbm(fetch_db, write_db, out, note=""):
    fetch, write = per_kernel(fetch_db), per_kernel(write_db)
    res = {}
    for k, (n, f) in fetch.items():
        w = write.get(k, (0, 0.0))[1]
        res[k] = dict(launches=n, fetch_size_kib_raw=f, write_size_kib=w, hbm_bytes_per_launch=(2 * f + w) * 1024.0)
    cal = res.get("dctfhe::k_affine")
    if cal:
        res["_calibration"] = dict(kernel="dctfhe::k_affine (reads N bytes, writes N bytes)", fetch_over_write=cal["fetch_size_kib_raw"] / cal["write_size_kib"])
    res["_note"] = note
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k in sorted([k for k in res if not k.startswith("_")], key=lambda k: -res[k]["hbm_bytes_per_launch"] * res[k]["launches"])[:8]:
        print("%-56s launches %4d  HBM %.2f GB/launch" % (k[-56:], res[k]["launches"], res[k]["hbm_bytes_per_launch"] / 1e9))
    print(res.get("_calibration"))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        hbm(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
