#!/usr/bin/env python3
"""evidence_summ_r03.py TAG PREFIX [KERNEL_SUBSTR ...]: gpurun_out/evidence_TAG (tools/evidence.sh) ->
profiles/r03/PREFIX_{kernel_stats.csv, pmc_*.csv, pmc_summary.json, bench.json}.

The step of the bf16 engine is several kernels (k_chain_fwd / _bwd / _wgrad + the thin-layer kernels), so the
summary is per STEP: every kernel launched between two Adam updates is summed.  HBM bytes follow
MI355X_MICROARCH.md: FETCH_SIZE is reported in KiB and counts 1/2 of wide coalesced reads on gfx950 (x2);
WRITE_SIZE (KiB) is exact for 16-byte-per-lane stores.  Kernels named on the command line get their own rows."""
import collections, csv, glob, json, os, shutil, sys

tag, prefix = sys.argv[1:3]
named = sys.argv[3:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "evidence_" + tag)
dst = os.path.join(root, "profiles", "r03")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"bench_{prefix}.json"))
def newest(pattern):
    """gpurun MERGES the box's gpurun_out into the local one: files of earlier evidence runs of the same tag stay behind
    under other PIDs.  Always take the newest."""
    return sorted(glob.glob(pattern), key=os.path.getmtime)[-1:]


st = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(dst, f"{prefix}_kernel_stats.csv"))
bench = json.load(open(os.path.join(src, "bench.json")))
pts = bench["config"]["points_per_gpu"]
names = {0: "fetch", 1: "write", 2: "sq", 3: "sq2"}
per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values (per dispatch)
n_adam = {}
for i in range(4):
    f = newest(os.path.join(src, f"pmc_{i}", "*", "*counter_collection.csv"))
    if not f:
        continue
    shutil.copy(f[0], os.path.join(dst, f"{prefix}_pmc_{names[i]}.csv"))
    adam = 0
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "k_adam" in k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"):
            adam += 1
    n_adam[i] = max(adam, 1)


def per_step(counter, pass_idx):
    """sum over every kernel of (all dispatches' counter values) / number of steps in that pass"""
    tot = 0.0
    for k, cs in per_kernel.items():
        tot += sum(cs.get(counter, []))
    return tot / n_adam.get(pass_idx, 1)


rd, wr = per_step("FETCH_SIZE", 0) * 1024 * 2, per_step("WRITE_SIZE", 1) * 1024
rows = {}
for sub in named:
    acc = collections.defaultdict(list)
    for k, cs in per_kernel.items():
        if sub in k:
            for c, v in cs.items():
                acc[c] += v
    if not acc:
        continue
    a = {c: sum(v) / len(v) for c, v in acc.items()}
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    rows[sub] = {
        "hbm_read_bytes": a.get("FETCH_SIZE", 0) * 2048, "hbm_write_bytes": a.get("WRITE_SIZE", 0) * 1024,
        "mfma_busy_fraction": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in a else None,
        "wait_any_fraction": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if "SQ_WAIT_ANY" in a else None,
        "insts_mfma": a.get("SQ_INSTS_MFMA"), "insts_valu_non_mfma": (a.get("SQ_INSTS_VALU", 0) - a.get("SQ_INSTS_MFMA", 0)) or None,
        "lds_bank_conflict_over_active": (a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"]) if a.get("SQ_LDS_IDX_ACTIVE") else None,
    }
out = {
    "what": f"one optimisation step of `bench.py {' '.join(bench.get('argv', []))}` ({prefix}); all kernels of the step summed",
    "command": "tools/evidence.sh: rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline ... (one pass per counter set)",
    "points_per_launch": pts,
    "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_step": rd + wr, "hbm_bytes_per_point": (rd + wr) / pts,
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE exact for 16-B/lane stores.",
    "ms_per_step": bench["ms_per_step"], "hbm_TBps": (rd + wr) / (bench["ms_per_step"] * 1e-3) / 1e12,
    "kernels": rows,
}
json.dump(out, open(os.path.join(dst, f"{prefix}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_step", "hbm_bytes_per_point", "ms_per_step", "hbm_TBps")}))
for k, v in rows.items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
