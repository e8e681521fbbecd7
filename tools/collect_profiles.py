"""Copy the summaries of one measurement pass from gpurun_out/ into profiles/ (the names bench.py and DESIGN.md cite):
    python tools/collect_profiles.py <tag>        e.g. r05k  (reads gpurun_out/<tag>_head, <tag>_patch, <tag>_ndprof, <tag>_bench_*.json,
                                                  <tag>_depth_gc.log / _sg.log when present)"""
import json
import pathlib
import re
import shutil
import sys

R = pathlib.Path(__file__).resolve().parents[1]
tag = sys.argv[1]
G = R / "gpurun_out"


def cp(src, dst):
    d = json.load(open(G / src))
    if "file" in d:
        d["file"] = "profiles/" + dst
    json.dump(d, open(R / "profiles" / dst, "w"), indent=1)
    print(dst, d.get("libpgx_sha256_16"), d.get("traffic_over_algorithmic"), d.get("per_factorisation"))
    return d.get("libpgx_sha256_16")


h = None
if (G / f"{tag}_head").exists():
    for k in ("spmv", "stspmv", "p2stspmv", "fsmooth"):
        h = cp(f"{tag}_head/{k}_pmc_traffic.json", f"r05_{k}_pmc_traffic.json")
    shutil.copy(G / f"{tag}_head/kernel_stats.csv", R / "profiles/r05_bench_2048_kernel_stats.csv")
    shutil.copy(G / f"{tag}_head/trace_by_level.txt", R / "profiles/r05_bench_2048_trace_by_level.txt")
if (G / f"{tag}_patch").exists():
    cp(f"{tag}_patch/patch_apply_pmc_traffic.json", "r05_patch_apply_pmc_traffic.json")
    cp(f"{tag}_patch/patch_edges_pmc_traffic.json", "r05_patch_edges_pmc_traffic.json")
if (G / f"{tag}_ndprof").exists():
    for t, name in (("ex06_1024", "r05_ex06_1024_kernel_stats.csv"), ("ex02_70", "r05_ex02_70cube_kernel_stats.csv")):
        h = cp(f"{tag}_ndprof/nd_traffic_{t}.json", f"r05_nd_traffic_{t}.json")
        shutil.copy(G / f"{tag}_ndprof/{t}_kernel_stats.csv", R / "profiles" / name)
for k, v in {"2048": "r05_bench_2048.json", "config3": "r05_bench_config3_p2_2048_one_gpu.json", "ex06": "r05_bench_ex06_1024.json",
             "ex02": "r05_bench_ex02_70cube.json"}.items():
    f = G / f"{tag}_bench_{k}.json"
    if f.exists():
        line = f.read_text().strip().splitlines()[-1]
        d = json.loads(line)
        (R / "profiles" / v).write_text(line + "\n")
        print(v, round(d["value"], 3), round(d["ms_per_step"], 1), round(d["roofline"]["frac"], 4))
gc, sg = G / f"{tag}_depth_gc.log", G / f"{tag}_depth_sg.log"
if gc.exists() and sg.exists() and h:
    def grab(p):
        t = p.read_text()
        return t[t.index("pgx_nd depth profile"):].strip()
    hdr = (R / "profiles/r05_nd_depth_profile.txt").read_text().splitlines()[:2]
    hdr[0] = re.sub(r"library [0-9a-f]{16}", f"library {h}", hdr[0])
    (R / "profiles/r05_nd_depth_profile.txt").write_text(
        "\n".join(hdr) + "\n\n## example 06 at 1024^2 (BASELINE config 4): tools/gc_scaling.py 1024\n" + grab(gc)
        + "\n\n## example 02 at 70^3 (BASELINE config 5): tools/sg_scaling.py 70\n" + grab(sg) + "\n")
    print("depth profile written, library", h)
