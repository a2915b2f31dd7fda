"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of scripts/pmc_traffic.sh.

    python scripts/pmc_traffic_summary.py gpurun_out/pmc_traffic_<dtype> profiles/<tag>_pmc_traffic_<dtype>.md profiles/traffic.json <dtype>

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  On gfx950 FETCH_SIZE reports half of a wide coalesced read
(MI355X_MICROARCH.md, HBM section; calibrated here on adam_kernel and relu-sized streaming kernels whose traffic is
known exactly), so reads are doubled; WRITE_SIZE is used as reported."""
import collections, csv, json, re, sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        a = agg[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    src, md, js = sys.argv[1:4]
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
    import glob
    f = load(glob.glob(src + "/FETCH_SIZE/**/*counter_collection.csv", recursive=True)[0], "FETCH_SIZE")
    w = load(glob.glob(src + "/WRITE_SIZE/**/*counter_collection.csv", recursive=True)[0], "WRITE_SIZE")
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/pmc_traffic.sh) over python bench.py --dtype <dtype> --steps 2 "
            "--warmup 1 --no-decode --no-cpu-baseline --no-extras; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads; "
            "check: adam_kernel must read 4 x and write 3 x the 124 MB parameter buffer), WRITE_SIZE as reported")
    rows = []
    for k in f:
        n, kib = f[k]
        wn, wkib = w.get(k, [0, 0.0])
        rows.append((k, n, kib / n, 2 * kib / n * 1024 / 1e6, wkib / max(wn, 1), wkib / max(wn, 1) * 1024 / 1e6))
    rows.sort(key=lambda r: -(r[3] + r[5]) * r[1])
    with open(md, "w") as o:
        o.write("# HBM traffic per kernel (PMC), training step, dtype %s\n\n%s\n\n" % (dtype, note))
        o.write("| kernel | launches | FETCH_SIZE KiB/launch (raw) | read MB/launch (x2) | WRITE_SIZE KiB/launch | write MB/launch |\n|---|---|---|---|---|---|\n")
        for r in rows:
            if r[0].startswith("fira::"):
                o.write("| `%s` | %d | %.1f | %.2f | %.1f | %.2f |\n" % r)

    def group(pred):
        n = sum(f[k][0] for k in f if pred(k))
        rd = sum(2 * f[k][1] for k in f if pred(k)) * 1024
        wr = sum(w[k][1] for k in w if pred(k)) * 1024
        return {"hbm_bytes_per_launch": (rd + wr) / max(n, 1), "read_bytes_per_launch": rd / max(n, 1),
                "write_bytes_per_launch": wr / max(n, 1), "launches": n}

    try:
        allj = json.load(open(js))
        if "gemm" in allj:                  # round-1 layout (fp32 only): move it under its dtype
            allj = {"f32": {k: allj[k] for k in ("gemm", "spmm") if k in allj}}
    except Exception:
        allj = {}
    allj[dtype] = {"gemm": group(lambda k: "gemm_" in k), "spmm": group(lambda k: "spmm_" in k),
                   "gcn": group(lambda k: "gcn_fused" in k),
                   "attention": group(lambda k: "attention_" in k),
                   "attention_cross_fwd": group(lambda k: "attention_fwd_kernel<12" in k),
                   "attention_cross_bwd": group(lambda k: "attention_bwd_kernel<12" in k)}
    import os
    allj[dtype]["commit"] = os.environ.get("ROCPD_COMMIT", "")
    allj["note"] = note
    json.dump(allj, open(js, "w"), indent=1)


if __name__ == "__main__":
    main()
