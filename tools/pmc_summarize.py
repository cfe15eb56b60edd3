"""summarise rocprofv3 --pmc counter_collection.csv files: per-kernel totals and per-wavefront-step values (diagnostic, not a test)"""
import csv, glob, json, sys, collections
def load(pattern, kernel_substr):
    tot = collections.defaultdict(float)
    for f in glob.glob(pattern):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
    return tot
if __name__ == "__main__":
    kern, waves, steps = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
    tot = {}
    for p in sys.argv[4:]:
        tot.update(load(p, kern))
    per = {k: v / (waves * steps) for k, v in tot.items()}
    print(json.dumps({"totals": tot, "per_wavefront_step": per}, indent=1))
