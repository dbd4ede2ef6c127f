"""Summarise rocprofv3 --pmc passes on the GPU box (the raw counter CSVs are too big to travel).

usage: pmc_summary.py OUT.json DIR [DIR ...]      each DIR = one `rocprofv3 --pmc X -d DIR` pass
Per kernel and counter: launches, mean value per launch, mean grid size.  FETCH_SIZE / WRITE_SIZE
are reported by rocprofv3 in KiB; per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE tallies 128-B requests at 64 B, so `hbm_read_bytes` = 2 x FETCH_SIZE x 1024; WRITE_SIZE
is taken as is (uncalibrated, the guide says so)."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    """Hash of the kernel sources (csrc/*.hip, *.h): identifies the build a set of counters belongs to."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "smplify-x-partial_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")) and f != "api.hip":      # (api.hip is the host layer: loops, launches, allocation -- not what the counters measure)
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    out_path, dirs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: [0, 0.0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    kn = row["Kernel_Name"].split("(")[0]
                    if "k_lbs_dense16<" in kn: kn = "k_lbs_dense16"      # (one kernel in three workgroup widths: launch_lbs_dense)
                    k = (kn, row["Counter_Name"])
                    a = acc[k]
                    a[0] += 1; a[1] += float(row["Counter_Value"]); a[2] += float(row.get("Grid_Size", 0) or 0)
    res = {}
    for (kern, ctr), (n, tot, grid) in sorted(acc.items()):
        e = res.setdefault(kern, {})
        e[ctr] = {"launches": n, "mean_per_launch": tot / n, "mean_grid_size": grid / n}
    for kern, e in res.items():
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * e["FETCH_SIZE"]["mean_per_launch"]
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_launch"] = 1024.0 * e["WRITE_SIZE"]["mean_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]["mean_per_launch"] > 0:
            # busy cycles are summed over the 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE comes back summed over
            # the 8 XCDs (1.28 M for a 62 us kernel = 8 x 160 k cycles), so one XCD's active time is /8
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (e["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0 * 256 * 4)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            hm = e["TCC_HIT_sum"]["mean_per_launch"] + e["TCC_MISS_sum"]["mean_per_launch"]
            e["l2_hit_rate"] = e["TCC_HIT_sum"]["mean_per_launch"] / hm if hm > 0 else None
        if "SQ_WAIT_ANY" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]["mean_per_launch"] > 0:
            e["wait_frac"] = e["SQ_WAIT_ANY"]["mean_per_launch"] / e["SQ_WAVE_CYCLES"]["mean_per_launch"]
    res["_meta"] = {"csrc_sha": csrc_sha()}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.startswith(("hbm", "mfma", "l2_", "wait_"))} for k, v in res.items() if k != "_meta"}))

if __name__ == "__main__":
    main()
