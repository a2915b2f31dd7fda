"""Per-kernel instruction mix from scripts/pmc_valu.sh: VALU / SALU instructions per wave, share of wave cycles spent
issuing vs waiting.   usage: pmc_valu_summary.py <counter_collection.csv>"""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        n[k] += 1
rows = []
for k, d in agg.items():
    w = max(d["SQ_WAVES"], 1.0)
    wc = max(d["SQ_WAVE_CYCLES"], 1.0)
    rows.append((d["SQ_WAVE_CYCLES"], k, n[k], w / max(n[k], 1), d["SQ_INSTS_VALU"] / w, d["SQ_INSTS_SALU"] / w,
                 d["SQ_INSTS_VMEM_RD"] / w, d["SQ_ACTIVE_INST_ANY"] / wc, d["SQ_WAIT_INST_ANY"] / wc))
rows.sort(reverse=True)
print("| kernel | launches | waves/launch | VALU/wave | SALU/wave | VMEM rd/wave | issuing | waiting |\n|---|---|---|---|---|---|---|---|")
for _, k, c, wpl, va, sa, vm, act, wait in rows:
    if k.startswith("fira::"):
        print("| `%s` | %d | %.0f | %.0f | %.0f | %.1f | %.0f %% | %.0f %% |" % (k[:70], c, wpl, va, sa, vm, 100 * act, 100 * wait))
