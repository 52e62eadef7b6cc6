"""Summarise rocprofv3 output directories into one JSON (run on the GPU box after the rocprofv3 passes).

    python tools/prof_collect.py OUT.json KERNEL_SUBSTRING ALG_BYTES_PER_LAUNCH dir1 [dir2 ...]

Each dir is the -d directory of one rocprofv3 run: `--kernel-trace --stats` (kernel_stats.csv) or `--pmc ...`
(counter_collection.csv).  Counters are averaged per launch of the named kernel, skipping the first `SKIP` (default 30, the
warm-up rollout of tools/prof_step.py).  HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes and are in kilobytes (x1024); the gfx950 FETCH_SIZE undercount correction (x2, calibrated on wide
coalesced streaming reads) is reported alongside as an upper bound."""
import csv, glob, json, os, sys

out_path, kname, alg_bytes = sys.argv[1], sys.argv[2], float(sys.argv[3])
skip = int(os.environ.get("SKIP", 30))
res = {"kernel_filter": kname, "skipped_warmup_launches": skip}
for d in sys.argv[4:]:
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        res.setdefault("kernel_stats", []).extend(
            [dict(name=r["Name"][:90], calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, pct=float(r["Percentage"])) for r in rows[:8]])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kname not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])      # sum over XCDs / instances of one dispatch
            res["vgpr"], res["sgpr"], res["lds_block"], res["scratch"] = r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]
        for c, per in acc.items():
            vals = [v for _, v in sorted(per.items(), key=lambda kv: int(kv[0]))][skip:]
            if vals:
                res.setdefault("counters", {})[c] = dict(launches=len(vals), mean_per_launch=sum(vals) / len(vals))
c = res.get("counters", {})
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    fk, wk = c["FETCH_SIZE"]["mean_per_launch"], c["WRITE_SIZE"]["mean_per_launch"]
    res["traffic"] = dict(FETCH_SIZE_KB_per_launch=fk, WRITE_SIZE_KB_per_launch=wk, hbm_bytes_per_launch_raw=(fk + wk) * 1024,
                          hbm_bytes_per_launch_fetch_x2=(2 * fk + wk) * 1024, algorithmic_bytes_per_launch=alg_bytes,
                          note="separate --pmc passes; KB units; fetch x2 = guide's gfx950 correction for wide coalesced reads, an upper bound here")
# derived: share of the 64 lanes that are active in the VALU instructions the kernel issues (rocprof's "VALUUtilization":
# SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)), and the kernel's VALU instructions per second against the measured issue peak
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] > 0:
    res["valu_active_lane_fraction"] = c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"])
if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
    res["wave_cycles_waiting_fraction"] = c["SQ_WAIT_ANY"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"]
g = lambda k: c[k]["mean_per_launch"] if k in c else None
if g("TCP_TOTAL_CACHE_ACCESSES_sum") and g("TCP_TCC_READ_REQ_sum") is not None:
    res["vector_l1_miss_fraction"] = (g("TCP_TCC_READ_REQ_sum") + (g("TCP_TCC_WRITE_REQ_sum") or 0.0)) / g("TCP_TOTAL_CACHE_ACCESSES_sum")
if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
    res["l1_to_l2_read_latency_cycles"] = g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")
if g("SQC_DCACHE_REQ") and g("SQC_DCACHE_MISSES") is not None:
    res["scalar_cache_miss_fraction"] = g("SQC_DCACHE_MISSES") / g("SQC_DCACHE_REQ")
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res.get("traffic", {})), [k["name"][:40] + f" {k['avg_us']:.1f}us x{k['calls']}" for k in res.get("kernel_stats", [])[:3]])
