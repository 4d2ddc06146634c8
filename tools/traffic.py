#!/usr/bin/env python3
"""Turn a tools/pmc.sh run (gpurun_out/<dir>/p*/.../*counter_collection.csv) into the HBM-traffic
figure bench.py reports as roofline.traffic.

  python tools/traffic.py gpurun_out/pmcN profiles/r01_traffic.json

Per MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from separate --pmc passes,
are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream.  The factor is
calibrated here on k_v2_carry<2>, which reads the a/b planes exactly once (known byte count)."""
import collections
import csv
import glob
import json
import sys


def main(src, dst):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(src + "/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    per = {}
    for k in tot:
        f = tot[k]["FETCH_SIZE"] / max(1, cnt[k]["FETCH_SIZE"])
        w = tot[k]["WRITE_SIZE"] / max(1, cnt[k]["WRITE_SIZE"])
        per[k] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes": (2.0 * f + w) * 1024.0}
    agg = [k for k in per if "v2::" in k]
    out = {
        "source": src,
        "method": "sum over the kernels of one smx_dev_aggregate_wta_pair call of "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024; FETCH_SIZE x2 = gfx950 correction, calibrated on "
                  "k_v2_carry<2> (reads a,b exactly once)",
        "aggregation_call_hbm_bytes": sum(per[k]["hbm_bytes"] for k in agg),
        "kernels": {k: per[k] for k in sorted(per)},
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: round(v["hbm_bytes"] / 1e6, 1) for k, v in per.items() if k in agg}, indent=1))
    print("aggregation call HBM MB:", round(out["aggregation_call_hbm_bytes"] / 1e6, 1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
