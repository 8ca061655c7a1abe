import csv, glob, sys, re
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f:
        print(d, "no stats"); continue
    print("==", d)
    tot = {}
    for r in csv.DictReader(open(f[0])):
        n = r["Name"]
        if "ncg::" not in n: continue
        short = re.sub(r"\(.*", "", n).replace("void ncg::", "").replace("ncg::", "")
        print("  %-52s calls %4s avg %9.1f us total %9.1f us" % (short[:52], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
