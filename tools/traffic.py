#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes of tools/gpu_suite.sh (gpurun_out/<tag>/pmc_FETCH_SIZE,
pmc_WRITE_SIZE) into the HBM-traffic figure bench.py reports as roofline.traffic.

  python tools/traffic.py gpurun_out/<tag> [profiles/rNN_traffic.json]

Per MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE come from separate --pmc passes
and are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced stream.  The x2 is
checked here on k_v5_wta / k_v4_wta2, which read the aggregated volumes exactly once (known byte count: the
bench's 2 x 1242 x 375 x 192 floats + the key planes)."""
import collections
import csv
import glob
import json
import os, sys

AGG_KERNELS = ("k_v4_", "k_v5_")      # every kernel of the fused aggregation call (either walker)
WALKERS = ("k_v5_walk", "k_v4_walk")


def main(src, dst=None):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(src + "/pmc_*/*/*counter_collection.csv"):
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
        per[k] = {"fetch_kib_raw": f, "write_kib": w, "hbm_bytes": (2.0 * f + w) * 1024.0,
                  "dispatches_seen": cnt[k]["FETCH_SIZE"]}
    agg = [k for k in per if any(a in k for a in AGG_KERNELS)]
    walk = [k for k in per if any(a in k for a in WALKERS)]
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (the GPU scripts run this from /tmp)
        import stereo_matching_cuda_amd as smx
        library = smx.lib().smx_version().decode()
    except Exception:                                   # noqa: BLE001 (dev tool: the tag is best effort)
        library = None
    out = {
        "source": src,
        "library": library,                              # bench.py quotes the file only for the build it was taken from
        "walker_hbm_bytes": sum(per[k]["hbm_bytes"] for k in walk),
        "method": "sum over the kernels of one smx_dev_aggregate_wta_pair call of "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch; FETCH_SIZE x2 = gfx950 correction "
                  "(MI355X_MICROARCH.md), checked on the WTA kernel (reads q exactly once)",
        "aggregation_call_hbm_bytes": sum(per[k]["hbm_bytes"] for k in agg),
        "kernels": {k: per[k] for k in sorted(per)},
    }
    if dst:
        json.dump(out, open(dst, "w"), indent=1)
    for k in sorted(agg):
        print(f"{k:60s} fetch_raw {per[k]['fetch_kib_raw']/1024:9.1f} MiB  write {per[k]['write_kib']/1024:9.1f} MiB"
              f"  hbm {per[k]['hbm_bytes']/1e6:9.1f} MB")
    print("aggregation call HBM MB:", round(out["aggregation_call_hbm_bytes"] / 1e6, 1), " walker:", round(out["walker_hbm_bytes"] / 1e6, 1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
