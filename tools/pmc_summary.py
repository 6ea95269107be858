"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) into
profiles/<tag>_pmc_hbm.json: per kernel, average HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB -- gfx950
reports exactly half of the bytes of a wide coalesced read stream (calibrated here on k_affine, which reads and
writes the same byte count: FETCH_SIZE is half of WRITE_SIZE).  usage: pmc_summary.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys

def per_kernel(d):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}

def main():
    fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
    out = {}
    for k, (n, f) in fetch.items():
        w = write.get(k, (0, 0.0))[1]
        out[k] = dict(launches=n, fetch_size_kib_raw=f, write_size_kib=w, hbm_bytes_per_launch=(2 * f + w) * 1024.0)
    cal = out.get("dctfhe::k_affine")
    if cal:
        out["_calibration"] = dict(kernel="dctfhe::k_affine (reads N bytes, writes N bytes)", fetch_over_write=cal["fetch_size_kib_raw"] / cal["write_size_kib"])
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k in sorted(out, key=lambda k: -out[k].get("hbm_bytes_per_launch", 0) * out[k].get("launches", 0))[:8]:
        if k.startswith("_"): continue
        print("%-44s launches %4d  HBM %.1f GB/launch" % (k[-44:], out[k]["launches"], out[k]["hbm_bytes_per_launch"] / 1e9))
    print(out.get("_calibration"))

if __name__ == "__main__":
    main()
