"""Per-kernel HBM traffic from the two rocprofv3 --pmc passes of scripts/pmc_traffic.sh.

    python scripts/pmc_traffic_summary.py gpurun_out/pmc_traffic profiles/<tag>_pmc_traffic.md profiles/traffic.json

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
    f = load(src + "/FETCH_SIZE/t_counter_collection.csv", "FETCH_SIZE")
    w = load(src + "/WRITE_SIZE/t_counter_collection.csv", "WRITE_SIZE")
    note = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/pmc_traffic.sh) over python bench.py --steps 2 "
            "--warmup 1 --no-decode --no-cpu-baseline; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads; "
            "check: adam_kernel must read 4 x and write 3 x the 124 MB parameter buffer), WRITE_SIZE as reported")
    rows = []
    for k in f:
        n, kib = f[k]
        wn, wkib = w.get(k, [0, 0.0])
        rows.append((k, n, kib / n, 2 * kib / n * 1024 / 1e6, wkib / max(wn, 1), wkib / max(wn, 1) * 1024 / 1e6))
    rows.sort(key=lambda r: -(r[3] + r[5]) * r[1])
    with open(md, "w") as o:
        o.write("# HBM traffic per kernel (PMC), batch 32 training step\n\n%s\n\n" % note)
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

    json.dump({"gemm": group(lambda k: "gemm_" in k), "spmm": group(lambda k: "spmm_" in k), "note": note},
              open(js, "w"), indent=1)


if __name__ == "__main__":
    main()
