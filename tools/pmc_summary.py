"""Per-launch counter values of one kernel from rocprofv3 --pmc output (csv).  usage: pmc_summary.py dir kernel_substr"""
import sys, csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[int(r["Dispatch_Id"])]["_t"] = (int(r.get("Start_Timestamp", 0) or 0), int(r.get("End_Timestamp", 0) or 0))
for d in sorted(rows):
    print(d, rows[d])
