import csv, glob, collections, sys
for v in sys.argv[1:]:
    f = glob.glob(f"gpurun_out/pmc_native_{v}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[1 if r["Kernel_Name"].startswith("void") else 0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in agg.items():
        print(v, k)
        for n, vv in sorted(c.items()):
            print(f"    {n:32s} {sum(vv)/len(vv):16.0f}")
