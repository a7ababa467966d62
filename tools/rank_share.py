"""Every rank's share of an N-way 8-row-stripe split, measured on ONE GPU in bench.py's timed shape (warm-up, then exactly K
steps handed over as launch sets of at most 64 frames, wall clock around enqueue + synchronize), plus the same K frames cut
into sub-sets that ride the pipeline slots (OKRT_OPT_PIPELINE).  The slowest rank is the N-GPU job's time before the gather.
usage: python tools/rank_share.py [--workload sponza] [--ranks 8] [--steps 20 128] [--repeats 5] [--out profiles/r04_rank_share.json]"""
import argparse, importlib, json, os, statistics, sys, time
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="sponza")
ap.add_argument("--ranks", type=int, default=8)
ap.add_argument("--steps", type=int, nargs="+", default=[20, 128])
ap.add_argument("--repeats", type=int, default=5)
ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 4, 5], help="sub-sets a K-step region is cut into (1 = bench.py's shape)")
ap.add_argument("--library-split", action="store_true", help="leave the cutting to okrt_render_frames (OKRT_OPT_PIPELINE default): only split 1 is issued")
ap.add_argument("--out", default="")
args = ap.parse_args()

builder, kwargs, W, H, desc = bench.WORKLOADS[args.workload]
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
ctx = rt.ctx


def timed(steps, split):
    """ms for `steps` frames issued as `split` equal sub-sets per 64-frame launch set, median of --repeats"""
    if not args.library_split:
        ctx.set_option(abi.OPT_PIPELINE, 32 if split > 1 else 0)  # (0: one set at a time, nothing is cut)
    vals = []
    for _ in range(args.repeats + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            m = min(64, steps - done)
            per = -(-m // split)
            k = 0
            while k < m:
                rt.renderFrames(min(per, m - k))
                k += min(per, m - k)
            done += m
        ctx.synchronize()
        vals.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(vals[1:])  # the first repeat allocates this shape's queues


table = {"workload": desc, "ranks": args.ranks, "stripe_rows": bench.STRIPE_ROWS, "repeats": args.repeats,
         "shape": "wall ms around enqueue + okrt_synchronize of K frames (launch sets of <= 64 frames), median; split s = every launch set issued as s "
                  "okrt_render_frames calls that take turns on the pipeline slots", "rows": []}
splits = [1] if args.library_split else args.splits
for r in [None] + list(range(args.ranks)):
    if r is None:
        ctx.set_tile(H, 0, 1)
    else:
        ctx.set_tile(bench.STRIPE_ROWS, r, args.ranks)
    rt.resetAccumulation()
    rt.renderFrames(5); rt.renderFrames(64); ctx.synchronize()
    row = {"rank": "full frame" if r is None else r, "rows": ctx.tile_rows()}
    for K in args.steps:
        for s in splits:
            row[f"K{K}_split{s}_ms"] = round(timed(K, s), 4)
    table["rows"].append(row)
    print(row, flush=True)
full = table["rows"][0]
summary = {}
for K in args.steps:
    for s in splits:
        key = f"K{K}_split{s}_ms"
        slowest = max(r[key] for r in table["rows"][1:])
        summary[key] = {"full_frame_ms": full[key], "slowest_rank_ms": slowest, "projected_speedup_before_gather": round(full[key] / slowest, 2)}
    best = min((max(r[f"K{K}_split{s}_ms"] for r in table["rows"][1:]), s) for s in splits)
    summary[f"K{K}_best"] = {"split": best[1], "slowest_rank_ms": best[0], "projected_speedup_vs_full_split1": round(full[f"K{K}_split1_ms"] / best[0], 2)}
table["summary"] = summary
print(json.dumps(summary, indent=1))
if args.out:
    json.dump(table, open(args.out, "w"), indent=1)
rt.close()
