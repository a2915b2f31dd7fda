"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into a per-kernel table (markdown).

    python scripts/rocpd_stats.py gpurun_out/prof_x/step_results.db [steps] > profiles/<name>.md
"""
import collections
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute("select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.accum_vgpr_count, s.group_segment_size "
                       "from %s d join %s s on d.kernel_id = s.id" % (disp, sym)).fetchall()
    try:        # busy time per HIP stream / hardware queue: which chain is the critical path of the step
        cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
        qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
        per_q = cur.execute("select %s, count(*), sum(end - start) / 1e3, min(start), max(end) from %s group by %s"
                            % (qcol, disp, qcol)).fetchall() if qcol else []
    except Exception:
        per_q = []
    agg = collections.OrderedDict()
    for name, st, en, vg, ag, lds in rows:
        short = re.sub(r"\(.*", "", name).replace(".kd", "")
        a = agg.setdefault(short, [0, 0.0, 1e30, 0.0, vg, ag, lds])
        d = (en - st) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls | calls/step | total us/step | %% | avg us | min us | max us | VGPR | AGPR | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %.1f | %.2f | %.1f | %.1f | %.1f | %s | %s | %s |" %
              (k[:120], v[0], v[0] / steps, v[1] / steps, 100 * v[1] / tot, v[1] / v[0], v[2], v[3], v[4], v[5], v[6]))
    print("\ntotal kernel time: %.1f us over %d dispatches (%.1f us/step over %d steps)" % (tot, len(rows), tot / steps, steps))
    # ROCPD_CLASSES_JSON=<file> ROCPD_DTYPE=f32|bf16 [ROCPD_BATCH ROCPD_COMMIT ROCPD_VALUE ROCPD_SOURCE]: per-class kernel time of this
    # trace, merged into <file>[dtype] -- bench.py quotes `roofline.rocprof` from it (the GEMM family's FLOP over the kernel
    # durations of this trace: what a reader of the table above recomputes)
    import json, os
    out = os.environ.get("ROCPD_CLASSES_JSON")
    if out:
        cls = [("gemm", ("gemm_", "wgrad_panel", "head_logits_x3")), ("gcn", ("gcn_fused",)), ("comb", ("comb_fused",)), ("attention", ("attention_",)), ("spmm", ("spmm_",)),
               ("copy", ("copy_score",)), ("head", ("head_loss",)), ("adam", ("adam_",))]
        acc = {c: [0, 0.0] for c, _ in cls}
        acc["rowops"] = [0, 0.0]
        for k, v in agg.items():
            if not ("fira" in k):
                continue
            for c, pats in cls:
                if any(p_ in k for p_ in pats):
                    acc[c][0] += v[0]; acc[c][1] += v[1]
                    break
            else:
                acc["rowops"][0] += v[0]; acc["rowops"][1] += v[1]
        try:
            allj = json.load(open(out))
        except Exception:
            allj = {}
        ent = {"%s_us_per_step" % c: a[1] / steps for c, a in acc.items()}
        ent.update({"%s_launches_per_step" % c: a[0] / steps for c, a in acc.items()})
        ent.update({"steps": steps, "total_kernel_us_per_step": tot / steps,
                    "batch": int(os.environ.get("ROCPD_BATCH", "0")), "commit": os.environ.get("ROCPD_COMMIT", ""),
                    "commits_per_s": float(os.environ.get("ROCPD_VALUE", "0") or 0),
                    "source": os.environ.get("ROCPD_SOURCE", "")})
        allj[os.environ.get("ROCPD_DTYPE", "f32")] = ent
        json.dump(allj, open(out, "w"), indent=1)
    if per_q and qcol:
        main_q = max(per_q, key=lambda r: r[1])[0]
        rows_q = cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id where d.%s = ? "
                             "order by d.start" % (disp, sym, qcol), (main_q,)).fetchall()
        aggq = collections.OrderedDict()
        gaps, prev_end = 0.0, None
        for name, st, en in rows_q:
            short = re.sub(r"\(.*", "", name).replace(".kd", "")
            a = aggq.setdefault(short, [0, 0.0])
            a[0] += 1; a[1] += (en - st) / 1e3
            if prev_end is not None and st > prev_end and st - prev_end < 50000:      # idle gaps < 50 us between kernels
                gaps += (st - prev_end) / 1e3
            prev_end = max(prev_end or 0, en)
        print("\nmain stream (%s): busy %.1f us/step, idle gaps between consecutive kernels %.1f us/step" %
              (main_q, sum(v[1] for v in aggq.values()) / steps, gaps / steps))
        print("\n| kernel on the main stream | calls/step | total us/step | avg us |")
        print("|---|---|---|---|")
        for k, v in sorted(aggq.items(), key=lambda kv: -kv[1][1])[:24]:
            print("| `%s` | %.1f | %.1f | %.1f |" % (k[:110], v[0] / steps, v[1] / steps, v[1] / v[0]))
    if qcol and len(sys.argv) > 3:
        # timeline of ONE step (the last but one adam_mb_kernel-delimited one): every dispatch on every stream, ordered by start
        allr = cur.execute("select s.kernel_name, d.start, d.end, d.%s, d.grid_size, d.workgroup_size from %s d join %s s on d.kernel_id = s.id "
                           "order by d.start" % (qcol, disp, sym)).fetchall() if "grid_size" in cols else \
            [r + (0, 0) for r in cur.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id "
                                             "order by d.start" % (qcol, disp, sym)).fetchall()]
        adam = [i for i, r in enumerate(allr) if "adam_mb" in r[0]]
        # (round 5: fira_train_step updates in up to three launches per step -- a step ends with the LAST of a run of Adam
        #  launches less than 300 us apart)
        adam = [i for n, i in enumerate(adam) if n + 1 == len(adam) or allr[adam[n + 1]][1] - allr[i][2] > 300e3]
        if len(adam) >= 3:
            lo, hi = adam[-3] + 1, adam[-2] + 1
            t0 = allr[lo][1]
            with open(sys.argv[3], "w") as f:
                f.write("| # | stream | start us | dur us | gap to prev on stream us | WGs | kernel |\n|---|---|---|---|---|---|---|\n")
                last_end = {}
                for i in range(lo, hi):
                    name, st, en, q, gs, ws = allr[i]
                    short = re.sub(r"\(.*", "", name).replace(".kd", "")
                    short = re.sub(r"^_ZN4fira\d+", "", short)[:70]
                    gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
                    last_end[q] = max(last_end.get(q, 0), en)
                    f.write("| %d | %s | %.1f | %.1f | %.1f | %s | `%s` |\n" % (i - lo, q, (st - t0) / 1e3, (en - st) / 1e3, gap,
                                                                             (gs // ws) if ws else "", short))
    if per_q:
        print("\n| stream / queue | dispatches | busy us/step | span ms |")
        print("|---|---|---|---|")
        for q, n, busy, t0, t1 in per_q:
            print("| %s | %d | %.1f | %.1f |" % (q, n, busy / steps, (t1 - t0) / 1e6))


if __name__ == "__main__":
    main()
