"""Copies the summaries of one scripts/round_profile.sh run from gpurun_out/ into profiles/ (tracked):
python scripts/collect_profiles.py <tag> <name>   e.g.  r02b r02_b"""
import csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
O, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(O, f"bench_{tag}.json"), os.path.join(P, f"{name}_cfg4_bench.json"))
shutil.copy(os.path.join(O, f"prof_{tag}", "t_kernel_stats.csv"), os.path.join(P, f"{name}_cfg4_kernel_stats.csv"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "kstats.py"), os.path.join(O, f"prof_{tag}", "t_kernel_trace.csv"),
                       os.path.join(P, f"{name}_cfg4_pass_kernels.csv")])
dm = os.path.join(O, f"prof_dm_{tag}", "t_kernel_stats.csv")
if os.path.exists(dm):
    shutil.copy(dm, os.path.join(P, f"{name}_dense_mode_kernel_stats.csv"))
os.makedirs(os.path.join(P, f"{name}_pmc"), exist_ok=True)
KEEP = ("chol_tile_kernel", "blk_", "rows_kernel", "zero_lower", "direct_kernel", "backsolve")
for kind in ("fetch", "write"):
    f = glob.glob(os.path.join(O, f"pmc_{kind}_{tag}", "*", "*counter_collection.csv"))
    if not f:
        continue
    rows = list(csv.DictReader(open(f[0])))
    big = {}
    for r in rows:
        if "chol_tile_kernel" in r["Kernel_Name"]:
            big[r["Dispatch_Id"]] = float(r["Counter_Value"])
    lim = 0.5 * max(big.values()) if big else 0
    with open(os.path.join(P, f"{name}_pmc", f"pmc_{kind}_dispatches.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Counter_Name", "Counter_Value"])
        for r in rows:
            if any(k in r["Kernel_Name"] for k in KEEP) and "slice_rows_kernel" not in r["Kernel_Name"] and not ("chol_tile_kernel" in r["Kernel_Name"] and float(r["Counter_Value"]) < lim):
                w.writerow([r["Dispatch_Id"], r["Kernel_Name"][:90], r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                            r["Counter_Name"], r["Counter_Value"]])
shutil.copy(os.path.join(O, f"pmc_traffic_{tag}.json"), os.path.join(P, f"{name}_pmc", "pmc_traffic.json")) if os.path.exists(os.path.join(O, f"pmc_traffic_{tag}.json")) else None
src = os.path.join(O, f"pmc_traffic_{tag}.json")
if os.path.exists(src):
    shutil.copy(src, os.path.join(P, "pmc_traffic.json"))      # the file bench.py reads its `traffic` from
for c in ("cfg2", "cfg3", "cfg4_local"):
    src = os.path.join(O, f"bench_{c}_{tag}.json")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{name}_{c}_bench.json"))
for f, dst in ((f"create_time_{tag}.log", f"{name}_create_time.log"), (f"exactN_cfg3b_{tag}.json", f"{name}_exactN_cfg3b.json"),
               (f"exactN_cfg4_{tag}.json", f"{name}_exactN_cfg4.json")):
    if os.path.exists(os.path.join(O, f)):
        shutil.copy(os.path.join(O, f), os.path.join(P, dst))
print("collected", name)
