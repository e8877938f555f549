import re, csv, glob, collections, sys, os
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
res = collections.defaultdict(dict)
for d in "abcdef":
    fs = glob.glob(f"{root}/pmc_{tag}_{d}/*/*_counter_collection.csv")
    if not fs: continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.defaultdict(float); disp = collections.Counter()
    for r in rows:
        m = re.search(r"k_[a-z_]+(<[^>]*>)?", r["Kernel_Name"])
        if not m: continue
        k = m.group(0).replace(", ", "_").replace("<", "_").replace(">", "")
        agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); disp[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / disp[(k, c)]
with open(f"{root}/pmc_{tag}_summary.csv", "w") as f:
    names = sorted({c for k in res for c in res[k]})
    f.write("kernel," + ",".join(names) + "\n")
    for k in res:
        f.write(k + "," + ",".join(str(int(res[k].get(c, 0))) for c in names) + "\n")
print(open(f"{root}/pmc_{tag}_summary.csv").read())
