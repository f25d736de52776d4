"""Per-kernel in-situ comparison of tools/prof_ab_libs.sh's two runs: total ms per kernel name, old vs new."""
import csv, glob, sys
def load(tag):
    f = glob.glob(f"gpurun_out/pab_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("old"), load("new")
rows = sorted(set(a) | set(b), key=lambda k: -(a.get(k, (0, 0))[1]))
ta = tb = 0.0
for k in rows[: int(sys.argv[1]) if len(sys.argv) > 1 else 30]:
    ca, xa = a.get(k, (0, 0.0)); cb, xb = b.get(k, (0, 0.0))
    print(f"{k[:90]:90s} {ca:5d} {xa:9.2f} | {cb:5d} {xb:9.2f}  {xb - xa:+8.2f} ms")
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
