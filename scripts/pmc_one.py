"""Mean value per launch of every counter in a rocprofv3 --pmc output directory, for kernels matching argv[2]."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
tot, cnt = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in tot:
    print(k, round(tot[k] / cnt[k]), "per launch over", cnt[k])
