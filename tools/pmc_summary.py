#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) per kernel, PER DISPATCH.

usage: pmc_summary.py <dir with one sub-dir per pass | an earlier summary.json> <out.json> [meta.json]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (MI355X_MICROARCH.md, HBM section); the parser's and the
finder's loads are narrow scattered reads, for which the guide's 2x "wide coalesced stream" correction does not apply --
values are reported uncorrected and flagged as such.

A bench run launches every kernel several times with different amounts of work (the timed batches, the small ratio /
round-trip encodes).  Dividing a summed counter by the number of dispatches mixes those (the round-3 review found the
"154 GB per launch" figure to be such a mean), so for every kernel this reports
  * `dispatch_rows`: one row per dispatch in program order -- grid size and every counter collected for it
    (the passes are joined by the dispatch's ordinal among the launches of that kernel);
  * `big_launches`: the dispatches whose grid is within 2 % of the largest one and which did at least half the work of the
    heaviest of those (the full timed batches; a persistent kernel keeps its grid for a partial pass) and
    `hbm_bytes_per_big_launch_uncorrected` = their mean (FETCH_SIZE + WRITE_SIZE) * 1024;
  * the plain sums over all dispatches (`sum_<counter>`), for cross-checks against the raw CSVs.
`_meta` carries the bench arguments the profile was taken with (corpus, preset, size), so that bench.py only quotes a
traffic figure for the same kernel variant AND workload.
"""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
meta = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and os.path.exists(sys.argv[3]) else {}


def short_name(k):
    s = k.replace("(anonymous namespace)::", "").replace("void ", "")
    if s.startswith("rocprim") or "rocprim::" in s.split("(")[0]:
        return "rocprim::" + s.split("rocprim::")[-1].split("<")[0].split("(")[0] + "(*)"
    return s.split("(")[0].strip()[:96]


# kernel -> counter -> list of (dispatch id, grid, value) in file order
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    seen = {}
    for r in csv.DictReader(open(f)):
        k = short_name(r["Kernel_Name"])
        c = r["Counter_Name"]
        key = (k, c, r["Dispatch_Id"])
        v = float(r["Counter_Value"])
        if key in seen:                      # one row per XCD / dimension: sum them
            seen[key][2] += v
            continue
        grid = int(float(r.get("Grid_Size", 0) or 0))
        ent = [int(r["Dispatch_Id"]), grid, v]
        seen[key] = ent
        rows[k][c].append(ent)

def summarise(table, ndisp, sums):
    e = {"dispatches": ndisp}
    e.update(sums)
    gmax = max((r.get("grid", 0) for r in table), default=0)
    big = [r for r in table if gmax and r.get("grid", 0) >= 0.98 * gmax]
    # a persistent kernel is launched with the same grid whatever its work (round 6: k_parse_pieces runs a partial iteration --
    # an eighth of the work -- and the full one with one grid): among the widest launches only those that did at least half the
    # work of the heaviest one are "big"
    for wc in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "FETCH_SIZE", "SQ_ACTIVE_INST_VALU"):
        if big and all(wc in r for r in big):
            wmax = max(r[wc] for r in big)
            big = [r for r in big if r[wc] >= 0.5 * wmax]
            break
    e["big_launches"] = len(big)
    e["big_launch_grid"] = gmax
    if big and all("FETCH_SIZE" in r and "WRITE_SIZE" in r for r in big):
        e["hbm_bytes_per_big_launch_uncorrected"] = sum((r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0 for r in big) / len(big)
        e["fetch_bytes_per_big_launch_uncorrected"] = sum(r["FETCH_SIZE"] * 1024.0 for r in big) / len(big)
    counters = set().union(*[set(r) for r in big]) - {"ordinal", "grid", "..."} if big else set()
    for c in sorted(counters):
        if c not in ("FETCH_SIZE", "WRITE_SIZE") and all(c in r for r in big):
            e["per_big_launch_" + c] = sum(r[c] for r in big) / len(big)
    # keep the per-dispatch rows of kernels that are launched a handful of times; for the rest (sort passes) the sums
    e["dispatch_rows"] = table if ndisp <= 64 else table[:8] + [{"...": ndisp - 16}] + table[-8:]
    return e


res = {"_meta": meta}
if src.endswith(".json"):
    # re-summarise an earlier summary from its per-dispatch rows (the raw counter CSVs of the SQ passes are too large to keep)
    old = json.load(open(src))
    res["_meta"] = old.get("_meta", meta)
    for k, e0 in old.items():
        if k == "_meta" or "dispatch_rows" not in e0:
            continue
        rows_ = [r for r in e0["dispatch_rows"] if "ordinal" in r]
        res[k] = summarise(rows_, e0.get("dispatches", len(rows_)), {c: v for c, v in e0.items() if c.startswith("sum_")})
else:
    for k, d in rows.items():
        ndisp = max(len(v) for v in d.values())
        table = []
        for i in range(ndisp):
            row = {"ordinal": i}
            for c, lst in d.items():
                if i < len(lst):
                    row["grid"] = lst[i][1]
                    row[c] = lst[i][2]
            table.append(row)
        res[k] = summarise(table, ndisp, {"sum_" + c: sum(x[2] for x in lst) for c, lst in d.items()})
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, e in sorted(((k, e) for k, e in res.items() if k != "_meta"), key=lambda kv: -kv[1].get("sum_FETCH_SIZE", 0)):
    print(k, {c: (int(v) if isinstance(v, float) else v) for c, v in e.items() if c != "dispatch_rows"})
