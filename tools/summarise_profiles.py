"""Summarise gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/ (tracked).  usage: python tools/summarise_profiles.py [tag=r03]"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
KERNEL = "phi_sort_kernel"
PHI_ALGORITHM = 6


def newest(pattern):
    """gpurun merges new files into gpurun_out/ without deleting older runs: keep only the most recent match."""
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1:]


def counters(sub):
    out = {}
    for f in newest(os.path.join(src, sub, "*", "*_counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                out.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def copy(name_in, name_out):
    p = os.path.join(src, name_in)
    if os.path.exists(p):
        if name_in.endswith(".txt") or name_in.endswith(".log"):
            lines = [ln for ln in open(p).read().splitlines() if "amdgpu.ids" not in ln and not ln.startswith("W2") and not ln.startswith("E20")]
            open(os.path.join(dst, name_out), "w").write("\n".join(lines) + "\n")
        else:
            shutil.copy(p, os.path.join(dst, name_out))


ks = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(ks, os.path.join(dst, tag + "_kernel_stats.csv"))
copy("bench_default.json", tag + "_bench.json")
copy("bench_sorted.json", tag + "_bench_sorted.json")
copy("trace_bench.json", tag + "_bench_under_tracer.json")
copy("dependent_timeline.txt", tag + "_dependent_timeline.txt")
copy("phi_ablation.json", tag + "_phi_ablation.json")
copy("mside_probe.txt", tag + "_mside_probe.txt")
copy("bcr_mfma_bench.txt", tag + "_bcr_mfma_levels.txt")
copy("phi_probe_10m.txt", tag + "_phi_probe_n10m.txt")
copy("phi_probe_1250k.txt", tag + "_phi_probe_n1250k.txt")
copy("vjp_probe.txt", tag + "_vjp_probe.txt")
copy("prior_dd_probe.txt", tag + "_prior_dd_probe.txt")
copy("dep_probe.txt", tag + "_dependent_step_ablation.txt")
copy("ahead_probe.txt", tag + "_ahead_probe.txt")
copy("reduce_bench.txt", tag + "_reduce_bench.txt")
copy("kron_probe_onesided.txt", tag + "_kron_probe_onesided.txt")
for sub in ("kron", "predict"):          # kernel-trace stats of tools/kron_probe.py / tools/predict_probe.py + the probes' own output
    fs = newest(os.path.join(src, sub, "*", "*_kernel_stats.csv"))
    if fs:
        shutil.copy(fs[0], os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, sub)))
        log = [ln for ln in open(os.path.join(src, sub + ".log")).read().splitlines() if "amdgpu.ids" not in ln and not ln.startswith("W2") and not ln.startswith("E20")]
        open(os.path.join(dst, "%s_%s_probe.txt" % (tag, sub)), "w").write("\n".join(log[-12:]) + "\n")
copy("kron_probe_untraced.txt", tag + "_kron_probe.txt")          # (the Kronecker probe's own timings: untraced; the trace above is tools/kron_trace.py)

# the bench runs its schedules in one process: split the Phi kernel's launches by grid size (256 workgroups = one step at a time and
# construction, fewer = the overlapped schedules) so that each average can be held against the matching figure of the bench line
kt = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if kt:
    by_grid = {}
    for row in csv.DictReader(open(kt[0])):
        if KERNEL in row["Kernel_Name"]:
            wgs = int(row["Grid_Size_X"]) // max(int(row["Workgroup_Size_X"]), 1)
            by_grid.setdefault(wgs, []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    json.dump({"kernel": KERNEL, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 3 --repeats 5 --no-cpu-baseline --no-extras",
               "launches_by_workgroups": {str(k): {"launches": len(v), "average_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
                                          for k, v in sorted(by_grid.items())},
               "note": "256 workgroups = one-step-at-a-time schedule (and warm-up / construction): compare with roofline.kernel_us; 240 = dependent and "
                       "independent-evaluation schedules (the launch shares the device with the chain workgroups)"},
              open(os.path.join(dst, tag + "_phi_kernel_by_schedule.json"), "w"), indent=1)
def kernel_of(sub):
    """name and average duration (us) of the Phi kernel in the kernel trace of a PMC pass"""
    names, durs = {}, []
    for f in newest(os.path.join(src, sub, "*", "*_kernel_trace.csv")):
        for row in csv.DictReader(open(f)):
            if KERNEL in row["Kernel_Name"]:
                names[row["Kernel_Name"].split("(")[0].replace("void asvgp::", "")] = 1
                durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    return " / ".join(names), (sum(durs) / len(durs) if durs else 0.0)


name = None
for order, sfx, sub_f, sub_w in (("unsorted", "", "fetch", "write"), ("sorted (time series)", "_sorted", "fetch_sorted", "write_sorted")):
    fetch, nf = counters(sub_f)
    write, nw = counters(sub_w)
    if "FETCH_SIZE" not in fetch or "WRITE_SIZE" not in write:
        continue
    kname, kus = kernel_of(sub_f)
    if not sfx:
        name = kname
    traffic = {
        "kernel": kname, "phi_algorithm": PHI_ALGORITHM, "input_order": order, "points_per_launch": 10_000_000,
        "FETCH_SIZE_KiB": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB": write["WRITE_SIZE"],
        "correction": "gfx950 tallies 128-B streaming reads at 64 B: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); units KiB",
        "hbm_bytes_per_launch": (2 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024,
        "algorithmic_bytes_per_launch": 160_000_000,
        "kernel_trace_average_us_under_pmc": kus,
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python3 tools/phi_pmc.py%s "
                   "(tools/collect_profiles.sh; launches averaged: %d / %d)" % (" with PHI_SORTED=1" if sfx else "", nf["FETCH_SIZE"], nw["WRITE_SIZE"]),
    }
    json.dump(traffic, open(os.path.join(dst, tag + "_phi_traffic" + sfx + ".json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
sq, _ = counters("sq")
sq2, _ = counters("sq2")
sq.update(sq2)
sqs, _ = counters("sq_sorted")
sqs2, _ = counters("sq2_sorted")
sqs.update(sqs2)
json.dump({"kernel": name, "unsorted": sq, "sorted": sqs,
           "note": "rocprofv3 --pmc, two passes of 8 SQ counters each per input order, averaged over the launches of tools/phi_pmc.py (chip totals; "
                   "PHI_SORTED=1 for the sorted input)"},
          open(os.path.join(dst, tag + "_phi_pmc_counters.json"), "w"), indent=1)
print(sq); print(sqs)
