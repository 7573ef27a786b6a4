#!/usr/bin/env python3
"""profiles/rNN_pmc.json from the rocprofv3 PMC passes of tools/profile_bench.sh (run on the GPU box).

  python tools/make_pmc_json.py <FETCH_SIZE results.db> <WRITE_SIZE results.db> <out.json> [build note
                                [<FETCH_SIZE results.db> <WRITE_SIZE results.db> of the batch-1 shape, B = 1 x 200 frames]]

Per kernel family (k_wn_layer*, k_flow_end4): launches, average FETCH_SIZE / WRITE_SIZE (KiB, as rocprofv3 reports them)
and HBM-side bytes per launch = 2 x FETCH (the gfx950 correction of MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies
128-byte requests at 64 bytes; re-checked every round on k_flow_end4, whose compulsory traffic is known) + WRITE.  The
summary records `kernel_source_id` (bench.kernel_source_id: the identity of csrc/facppg_wg.hip it was taken from), which
bench.py compares with the source it runs from before quoting `roofline.traffic`.
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
                     "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    return {n: (k, v, d / 1e3) for n, k, v, d in rows}


def family(table, prefix):
    items = [(n, v) for n, v in table.items() if n.startswith(prefix)]
    calls = sum(v[0] for _, v in items)
    if not calls:
        return None
    return {"launches": calls, "avg": sum(v[0] * v[1] for _, v in items) / calls, "avg_us": sum(v[0] * v[2] for _, v in items) / calls,
            "per_instantiation": {n[:120]: {"launches": v[0], "avg": v[1], "avg_us": v[2]} for n, v in sorted(items)}}


def main():
    import bench
    fetch_db, write_db, out_path = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    positions = bench.BATCH * bench.FRAMES * bench.HOP // 8
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh) -- "
                     "python bench.py --workload infer --no-cpu-baseline --steps 1 --warmup 0 (k_wn_layer_b1_t200, k_cond_seed: python tools/seeded_workload.py, the headline utterance's vocoder launches)",
           "correction": "gfx950 FETCH_SIZE reports 1/2 of coalesced reads (MI355X_MICROARCH.md, HBM section) -> fetch doubled; checked in "
                         "this same run on k_flow_end4, whose compulsory traffic is known (see its entry); Infinity-Cache hits are "
                         "included, so this is L2-miss traffic, an upper bound on HBM bytes"}
    b1 = (per_kernel(sys.argv[5], "FETCH_SIZE"), per_kernel(sys.argv[6], "WRITE_SIZE")) if len(sys.argv) > 6 else None
    wn = "void facppg::(anonymous namespace)::k_wn_layer"
    entries = [("k_wn_layer", wn, positions * 1984.0, fetch, write), ("k_flow_end4", "void facppg::(anonymous namespace)::k_flow_end4", None, fetch, write)]
    if b1 is not None:      # the metric's batch-1 utterance: python bench.py --workload infer --infer-batch 1 --infer-frames 200
        # (tools/seeded_workload.py: the streamed utterance's launches -- 160 seeded + 40 unseeded frames; algorithmic bytes per
        #  position as for k_wn_layer + the seeds a seeded tile reads, 512 rows x 4 B per position; k_cond_seed: one pass)
        entries.append(("k_wn_layer_b1_t200", wn + "_mixed", 200 * bench.HOP // 8 * 1984.0 + 160 * bench.HOP // 8 * 2048.0, b1[0], b1[1]))
        entries.append(("k_cond_seed", "void facppg::(anonymous namespace)::k_cond_seed", None, b1[0], b1[1]))
    for key, prefix, algo, ft, wt in entries:
        f, w = family(ft, prefix), family(wt, prefix)
        if f is None or w is None:
            continue
        hbm = (2.0 * f["avg"] + w["avg"]) * 1024.0
        e = {"launches": f["launches"], "FETCH_SIZE_KiB_avg": f["avg"], "WRITE_SIZE_KiB_avg": w["avg"], "avg_us": f["avg_us"],
             "hbm_bytes_per_launch": hbm, "achieved_TBps": hbm / (f["avg_us"] * 1e-6) / 1e12,
             "per_instantiation": {n: {"launches": v["launches"], "FETCH_SIZE_KiB": v["avg"],
                                       "WRITE_SIZE_KiB": w["per_instantiation"].get(n, {}).get("avg"), "avg_us": v["avg_us"]}
                                   for n, v in f["per_instantiation"].items()}}
        if algo:
            e["algorithmic_bytes_per_launch"] = algo
        e["kernel_source_id"] = bench.kernel_source_id()
        e["build"] = note
        out[key] = e
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: (v.get("hbm_bytes_per_launch"), v.get("avg_us")) for k, v in out.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
