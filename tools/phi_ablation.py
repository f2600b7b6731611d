"""profiles/r04_phi_ablation.json: the stand-alone Phi harness (tools/micro/phi_sort_bench, built by tools/micro/build_phi_sort_bench.sh) on
unsorted / sorted / clustered input - read-only stream ceilings, the compile-time stage ablation of phi_sort_kernel (loads + cell search /
+ rank atomics / + scan / + scatter / + owners' register moments / full kernel with epilogue) and the per-phase cycle stamps.
usage: python tools/phi_ablation.py out.json [N=10000000]"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "micro", "bin", "phi_sort_bench")
out_path = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
res = {"harness": "tools/micro/phi_sort_bench.hip (same phi_sort.hpp as the library)", "N": N, "M": 2048, "order": 4,
       "unit": "us per launch (20 launches between two events)", "inputs": {}}
num = r"([0-9.]+)"
for dist, name in ((0, "unsorted"), (1, "sorted"), (2, "clustered (N(0.5, 0.08))")):
    txt = subprocess.run([exe, str(N), str(dist), "1" if dist == 0 else "0"], capture_output=True, text=True, timeout=280).stdout
    print(txt, flush=True)
    d = {}
    m = re.search(r"nt depth 1/2/4/8 = %s / %s / %s / %s us; plain depth 4/8 = %s / %s us; 512 WGs nt depth 4 = %s us" % ((num,) * 7), txt)
    if m:
        v = [float(g) for g in m.groups()]
        d["stream_ceiling_us"] = {"nt_depth_1": v[0], "nt_depth_2": v[1], "nt_depth_4": v[2], "nt_depth_8": v[3], "plain_depth_4": v[4],
                                  "plain_depth_8": v[5], "nt_depth_4_512_workgroups": v[6]}
    for m in re.finditer(r"sort TP=(\d)\s+loads\+search %s \| \+rank %s \| \+scan %s \| \+scatter %s \| \+owners %s \| full %s us.*?late prefetch: \+owners %s full %s\]" % ((num,) * 8), txt):
        g = m.groups()
        d["points_per_thread_and_tile_%s" % g[0]] = {"loads_and_search": float(g[1]), "plus_rank": float(g[2]), "plus_scan": float(g[3]),
                                                     "plus_scatter": float(g[4]), "plus_owners": float(g[5]), "full": float(g[6]),
                                                     "late_prefetch_plus_owners": float(g[7]), "late_prefetch_full (product: TP=6)": float(g[8])}
    m = re.search(r"product \(late prefetch\): TP=6 without / with the time-series front loop: %s / %s us;  TP=4: %s / %s us" % ((num,) * 4), txt)
    if m:
        v = [float(g) for g in m.groups()]
        d["product_kernels_us"] = {"TP6_plain (unsorted default)": v[0], "TP6_with_front_loop": v[1], "TP4_plain": v[2], "TP4_with_front_loop (time-series default)": v[3]}
    d["phase_stamps"] = [ln.strip() for ln in txt.splitlines() if "cycles over all tiles" in ln]
    chk = [ln.strip() for ln in txt.splitlines() if ln.startswith("check")]
    if chk:
        d["check_vs_host"] = chk[0]
    res["inputs"][name] = d
json.dump(res, open(out_path, "w"), indent=1)
print("wrote", out_path)
